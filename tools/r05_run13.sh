cd $GRAFT_REPO_ROOT
timeout 400 python bench.py --force_dist --steps 8 --warmup 6 --global_batch 3 --no_cpu_baseline --no_kernel_timing 2>&1 | tail -30 | cut -c1-600
