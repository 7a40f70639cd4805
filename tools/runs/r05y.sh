timeout 280 python tools/runs/r05y.py 2>&1 | grep -v amdgpu.ids
rocm-smi --showmeminfo vram 2>/dev/null | grep -i "used\|total" | head -4
