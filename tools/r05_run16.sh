cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
export MASTER_ADDR=127.0.0.1
for w in 2 4; do
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=$w --master-addr 127.0.0.1 --master-port $((29800+w)) tools/ipc_exchange_probe.py gpurun_out/r05c/ipc_probe_w$w.json 2>&1 | grep -v "amdgpu.ids\|Warning\|warn" | tail -3
done
timeout 2400 python -m pytest tests/test_ddp_gpu.py -q -m gpu 2>&1 | tail -5
