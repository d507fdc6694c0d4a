for cam in 3 12 58 88 38; do
  for ph in 1 0; do
    r=$(VOXE_TILE_PHASES=$ph python bench.py --camera $cam --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step'])")
    echo "cam $cam phases=$ph: $r"
  done
done
for img in 100 200 266; do
  for ph in 1 0; do
    r=$(VOXE_TILE_PHASES=$ph python bench.py --image $img --camera 3 --steps 30 --warmup 8 --no-cpu-baseline --no-gpu-baseline --no-secondary 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); p=d['roofline']['phases_ms']; print(p['fwd'], p['bwd'], d['ms_per_step'], round(d['value']/1e6,1))")
    echo "image $img phases=$ph: $r"
  done
done
