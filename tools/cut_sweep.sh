for p in "8,4,32,8,512" "4,8,16,16,256" "2,16,8,32,128" "2,32,4,64,64" "1,64,2,128,32" "2,16,16,32,64" "4,16,4,64,16"; do
  echo "== $p"; VOXE_CUT_PARAMS=$p VOXE_REFINE_VERBOSE=1 timeout 300 python tools/refine_bench.py 160 2>&1 | grep -E "round|HIP|equal" | tail -4
done
