# round 6: 30 runs of the mailbox lockstep harness at 4 processes on one GPU (60 s timeout), then runs with a timeout short enough to
# fire under time slicing (what the old protocol did with it is in profiles/r05_ipc_exchange.md): results must be NaN + reported, never
# silently different.   usage: bash tools/r6_lockstep.sh [runs]
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out/r6_lockstep.txt; : > $O
export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
RUNS=${1:-30}
for i in $(seq 1 $RUNS); do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port $((29800 + i)) tools/ipc_lockstep_harness.py 400 60000 2>&1 | grep LOCKSTEP_RUN >> $O || echo "run $i: no result (rc $?)" >> $O
done
echo "# a rank that stalls for 6 s in front of exchange 120 with a 2 s timeout (a 200 ms one does not even pass the attach-time self-test under four time-sliced processes): the peers time out, every rank is poisoned: nan=true, first_diff=120, and the run ends in seconds (not 280 x 2 s)" >> $O
for i in 1 2 3; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port $((29900 + i)) tools/ipc_lockstep_harness.py 400 2000 2:120:6.0 2>&1 | grep LOCKSTEP_RUN >> $O || echo "stall run $i: no result (rc $?)" >> $O
done
cat $O
