python -m pytest tests -m gpu -x -q 2>&1 | tail -1
for bsz in 256 384 512 256 512; do python bench.py --no-cpu-baseline --steps 10 --warmup 2 --batch $bsz 2>/dev/null | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('batch', $bsz, d['value'], d['ms_per_step'])"; done
