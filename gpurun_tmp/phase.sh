python -m pytest tests/test_gpu_parity.py tests/test_gpu_mapping.py -m gpu -x -q 2>&1 | tail -1
for v in A B A B; do ALOAM_MI355X_LIB=$PWD/a-loam_amd/lib/libaloam_$v.so python bench.py --no-cpu-baseline --steps 10 --warmup 2 --batch 256 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['roofline']['kernels_ms_per_step']; print('$v', d['value'], d['ms_per_step'], 'ring_features', k['k_ring_features'])"; done
