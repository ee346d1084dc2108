GSPLAT_HIP_LIBRARY=$PWD/ab_variants/lib_group512.so python -m pytest tests/test_gpu_parity.py -x -q -k "two_level or fusion or tile_list or views or band or morton or staged" 2>&1 | tail -2
tools/ab_bench.sh 4 tree ab_variants/lib_group512.so 2>/dev/null | grep value
for r in 1 2; do for lib in tree ab_variants/lib_group512.so; do
 if [ $lib = tree ]; then v=$(python bench.py --splats 5000000 --width 3840 --height 2160 --steps 64 --warmup 8 --cpu-frames 0 --no-extra 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*'); else v=$(GSPLAT_HIP_LIBRARY=$PWD/$lib python bench.py --splats 5000000 --width 3840 --height 2160 --steps 64 --warmup 8 --cpu-frames 0 --no-extra 2>/dev/null | tail -1 | grep -o '"value": [0-9.]*'); fi; echo "C5 $lib $v"; done; done
