R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out/r6_lw; mkdir -p $O
python tools/lw_probe.py 3,24 > $O/base.txt 2>&1
UCD_CONV_LW_NL=8 python tools/lw_probe.py 3,6 > $O/nl8.txt 2>&1
UCD_CONV_LW_NL=8 UCD_CONV_LW_PF=1 python tools/lw_probe.py 3,6 > $O/nl8_pf1.txt 2>&1
UCD_CONV_BN64_TILES=0 python tools/lw_probe.py 3 > $O/bn128.txt 2>&1
tail -n 30 $O/*.txt
