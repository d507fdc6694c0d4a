export TMPDIR=/tmp
O=gpurun_out/r04_check7; mkdir -p $O
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -12 | tee $O/pytest_gpu.log
timeout 300 python tools/grid_pass_bench.py 2>&1 | tee $O/grid_passes.txt | tail -14
