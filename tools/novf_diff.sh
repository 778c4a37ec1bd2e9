#!/bin/bash
# The product library against the same sources built WITHOUT -amdgpu-mfma-vgpr-form (make -C ratilqr.jl_amd/csrc novf), bit by bit, on the
# random problems of tools/soak_parity.py: the default paths (mixed widths, rounds), the time-parallel latency kernel (SOAK_E=1; one and two
# workgroups per sample), the register forms of general sizes.  Through gpurun:  bash tools/novf_diff.sh
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=$R/gpurun_out/r06/novf; mkdir -p $O
cd $R
N=${SOAK_N:-1500}
rc=0
for cfg in "default:" "e1:SOAK_E=1" "e1solo:SOAK_E=1 RATILQR_PSW_DUO=0" "wide:SOAK_WIDE=1" "wide16:SOAK_WIDE=16"; do
    name=${cfg%%:*}; envs=${cfg#*:}
    n=$N; [ $name = wide ] && n=$((N / 4)); [ $name = wide16 ] && n=$((N / 2))
    env $envs SOAK_N=$n SOAK_DUMP=$O/$name.vf.npz timeout 900 python tools/soak_parity.py > $O/$name.vf.log 2>&1
    env $envs SOAK_N=$n SOAK_DUMP=$O/$name.novf.npz RATILQR_SO=$R/ratilqr.jl_amd/csrc/libratilqr_hip_novf.so timeout 900 python tools/soak_parity.py > $O/$name.novf.log 2>&1
    echo "$name ($envs): $(tail -n 1 $O/$name.vf.log) | novf: $(tail -n 1 $O/$name.novf.log)"
    python tools/novf_diff.py $O/$name.vf.npz $O/$name.novf.npz || rc=1
done
timeout 600 python tools/novf_batch.py $O/batch.vf.npz > $O/batch.vf.log 2>&1
RATILQR_SO=$R/ratilqr.jl_amd/csrc/libratilqr_hip_novf.so timeout 600 python tools/novf_batch.py $O/batch.novf.npz > $O/batch.novf.log 2>&1
echo "large batches, CE / NM solves, W(k): $(tail -n 1 $O/batch.vf.log) | novf: $(tail -n 1 $O/batch.novf.log)"
python tools/novf_diff.py $O/batch.vf.npz $O/batch.novf.npz || rc=1
rm -f $O/*.npz
exit $rc
