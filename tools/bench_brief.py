"""Print the scalars and the named legs of a bench.py JSON line:  python tools/bench_brief.py FILE [leg ...]"""
import json, sys
r = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print({k: (round(v, 4) if isinstance(v, float) else v) for k, v in r.items() if not isinstance(v, (dict, list, str))})
for leg in sys.argv[2:]:
    print(leg, json.dumps(r.get(leg), indent=1))
