for lib in "" "variants/libvoxe_hip_tex1.so"; do
  echo "== lib: ${lib:-shipped (packed texels from L2, 4 blocks per CU)}"
  for args in "--image 181 --ray-order random" "--image 256 --ray-order random" "--image 400 --ray-order random"; do
    VOXE_HIP_LIB=$lib python bench.py $args --no-cpu-baseline --no-gpu-baseline --no-secondary --steps 20 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print('   $args', p['fwd'], p['bwd'], d['ms_per_step'])"
  done
  VOXE_HIP_LIB=$lib python tools/recon_bench.py 30 2>/dev/null | tail -1
  VOXE_HIP_LIB=$lib python tools/attn_bench.py 2>/dev/null | tail -3
done
