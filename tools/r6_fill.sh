R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; cd $R; O=gpurun_out/r6_fill.txt; : > $O
for grid in 104 256; do
  for mode in 0 1 2 3; do
    for waves in 4 8; do
      for pitch in 512 2048 4608; do
        ./build_probe/fill_rate $mode $waves $pitch $grid >> $O 2>&1
      done
    done
  done
done
./build_probe/fill_rate 0 12 512 104 >> $O; ./build_probe/fill_rate 0 2 512 104 >> $O; ./build_probe/fill_rate 0 1 512 104 >> $O
./build_probe/fill_rate 1 12 512 104 >> $O; ./build_probe/fill_rate 1 2 512 104 >> $O; ./build_probe/fill_rate 1 1 512 104 >> $O
cat $O
