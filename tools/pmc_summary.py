"""Per-kernel summary of rocprofv3 --pmc counter_collection.csv files (one counter per pass, as the MI355X guide prescribes).
usage: python tools/pmc_summary.py FETCH=<counter_collection.csv> WRITE=<counter_collection.csv> [KERNEL_SUBSTR]
Prints launches, mean and max of each counter per kernel; for FETCH_SIZE applies the gfx950 correction (x2: 128-B requests
are tallied at 64 B) and reports the corrected traffic (MB) of the largest launch."""
import csv, sys, collections


def load(path):
    per = collections.defaultdict(list)
    with open(path) as f:
        for row in csv.DictReader(f):
            per[row["Kernel_Name"]].append(float(row["Counter_Value"]))
    return per


def main():
    files = dict(a.split("=", 1) for a in sys.argv[1:] if "=" in a)
    filt = [a for a in sys.argv[1:] if "=" not in a]
    fetch, write = load(files["FETCH"]), load(files["WRITE"])
    print("| kernel | launches | FETCH_SIZE mean KB | FETCH_SIZE max KB | WRITE_SIZE mean KB | WRITE_SIZE max KB | corrected traffic of the largest launch (MB) |")
    print("|---|---|---|---|---|---|---|")
    for k in sorted(set(fetch) | set(write)):
        if filt and not any(s in k for s in filt):
            continue
        fv, wv = fetch.get(k, [0.0]), write.get(k, [0.0])
        traffic = (2.0 * max(fv) + max(wv)) * 1024 / 1e6
        print(f"| `{k[:70]}` | {len(fv)} | {sum(fv)/len(fv):.1f} | {max(fv):.1f} | {sum(wv)/len(wv):.1f} | {max(wv):.1f} | {traffic:.1f} |")


if __name__ == "__main__":
    main()
