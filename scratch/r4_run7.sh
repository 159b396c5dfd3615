for r in 1 2; do TAG=hotmap python scratch/scan_ab.py 8 32; TAG=old MOCAP_SCAN_HOTMAP=0 python scratch/scan_ab.py 8 32; done 2>&1 | grep -v Warning
echo "--- wide-tile kernel occupancy (depth 1; filter = box + wide tiles)"
for m in 8 32; do
  bash scratch/ab_lib_env.sh "scratch/libs/base.so" "--depth 1 --markers $m" "MOCAP_WIDE_BLOCKS_PER_CU=2" "MOCAP_WIDE_BLOCKS_PER_CU=3" "MOCAP_WIDE_BLOCKS_PER_CU=3 MOCAP_WIDE_QUADS=34,34" | head -3
  bash scratch/ab_lib_env.sh "scratch/libs/rows_w4.so" "--depth 1 --markers $m" "MOCAP_WIDE_BLOCKS_PER_CU=4" "MOCAP_WIDE_BLOCKS_PER_CU=4 MOCAP_WIDE_QUADS=34,34" | head -2
done
echo "--- pipeline"
bash scratch/ab_lib_env.sh "scratch/libs/lean.so" "" "A=1" "MOCAP_SCAN_HOTMAP=0" "MOCAP_SCAN_BLOCKS_PER_CU=4 MOCAP_SCAN_SERIAL=1" "MOCAP_SCAN_BLOCKS_PER_CU=4 MOCAP_SCAN_SERIAL=1 MOCAP_SCAN_HOTMAP=0" "MOCAP_SCAN_SERIAL=1"
