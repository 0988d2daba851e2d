PMX_ALIGN_PROF=1 python tools/real_reads.py 1 2>&1 | grep -v "^step\|amdgpu.ids" | tail -8
