for cfg in "--splats 5000000" "--splats 5000000 --width 3840 --height 2160" "--splats 2000000"; do
for a in "" "--depth-layers 2" "--depth-layers 3" "--depth-layers 2 --layer-fractions 0.08" "--depth-layers 3 --layer-fractions 0.04,0.15"; do
  v=$(python bench.py $cfg --steps 128 --warmup 16 --cpu-frames 0 --no-extra $a 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'])")
  echo "[$cfg] [$a] -> $v"
done; done
