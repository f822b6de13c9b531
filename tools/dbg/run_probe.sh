cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -shared -fPIC -Wno-unused-value -Wno-unused-result tools/lds_polluter.hip -o tools/liblds_polluter.so 2>/dev/null
python3 - <<'PY'
import ctypes, time
import torch; torch.cuda.init(); torch.zeros(1, device="cuda")
L = ctypes.CDLL("tools/liblds_polluter.so")
L.pollute_probe.restype = ctypes.c_double
print("before the polluter:", [round(L.pollute_probe(0x7fc00001, 512), 4) for _ in range(3)])
L.pollute_start(0, 0x7fc00001)
time.sleep(0.5)
print("polluter running:   ", [round(L.pollute_probe(0x7fc00001, 512), 4) for _ in range(5)])
L.pollute_stop()
print("after it stopped:   ", [round(L.pollute_probe(0x7fc00001, 512), 4) for _ in range(3)])
PY
