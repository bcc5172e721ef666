#!/bin/bash
# stream priorities of the side streams (diagnostic)
R=$GRAFT_REPO_ROOT; cd $R
python - <<'PY'
import ctypes
h = ctypes.CDLL("libamdhip64.so")
lo, hi = ctypes.c_int(), ctypes.c_int()
print("hipDeviceGetStreamPriorityRange rc", h.hipDeviceGetStreamPriorityRange(ctypes.byref(lo), ctypes.byref(hi)), "least", lo.value, "greatest", hi.value)
PY
for i in 1 2; do for p in 0 1 -1; do echo -n "DYCON_SIDE_PRIORITY=$p  "; DYCON_SIDE_PRIORITY=$p timeout -k 10 200 bash tools/variant_bench.sh dycon_paper_replication_amd/libdycon_hip.so; done; done 2>&1 | grep -v amdgpu.ids | tee gpurun_out/side_priority.txt
