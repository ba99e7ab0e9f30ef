cd $GRAFT_REPO_ROOT; O=gpurun_out/r05a; mkdir -p $O
for dbg in 0 1 2 3; do echo "UCD_RA_DEBUG=$dbg"; UCD_RA_DEBUG=$dbg UCD_CONV_RA=1 timeout 300 python tools/conv_ra_probe.py 2>&1 | grep "images 24"; done > $O/conv_ra_debug.txt 2>&1; cat $O/conv_ra_debug.txt
