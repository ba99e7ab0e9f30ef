cd $GRAFT_REPO_ROOT; export HSA_ENABLE_IPC_MODE_LEGACY=0 MASTER_ADDR=127.0.0.1
for i in 1 2 3; do
  timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=4 --master-addr 127.0.0.1 --master-port $((29900 + i)) tools/ipc_lockstep_harness.py 400 2000 2:120:6.0 2>&1 | grep "LOCKSTEP_RUN\|Error" | head -5
done
