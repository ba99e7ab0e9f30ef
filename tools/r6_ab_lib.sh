# same-box A/B of two builds of libucd_hip.so (build_probe/libucd_hip_old.so against the in-tree one), alternating; usage: bash tools/r6_ab_lib.sh [batch] [reps]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out/r6_ab_lib.txt; B=${1:-24}; N=${2:-3}
cp ucd_amd/libucd_hip.so /tmp/lib_new.so
for i in $(seq 1 $N); do
  for v in old new; do
    if [ $v = old ]; then cp build_probe/libucd_hip_old.so ucd_amd/libucd_hip.so; else cp /tmp/lib_new.so ucd_amd/libucd_hip.so; fi
    python bench.py --global_batch $B --steps 20 --warmup 5 --no_cpu_baseline --no_kernel_timing 2>/dev/null | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$v batch $B: %.2f ms/step %.1f img/s' % (d['ms_per_step'], d['value']))" | tee -a $O
  done
done
cp /tmp/lib_new.so ucd_amd/libucd_hip.so
