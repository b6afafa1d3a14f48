for o in 0 1 2 3 4 6 12 16 24; do echo -n "tile_order $o: "; AVEX_AMD_GEMM_TILE_ORDER=$o python scripts/gemm_forms.py --plain --shapes out --iters 40 2>&1 | grep -v amdgpu | tail -1; done
for o in 0 4 12 16; do echo -n "qkv tile_order $o: "; AVEX_AMD_GEMM_TILE_ORDER=$o python scripts/gemm_forms.py --plain --shapes qkv --iters 40 2>&1 | grep -v amdgpu | tail -1; done
