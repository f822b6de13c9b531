import os, sys, time, tempfile, ctypes as C
sys.path.insert(0, os.getcwd())
from gauspcc_amd import _lib
from concurrent.futures import ThreadPoolExecutor
L=_lib.lib()
blob=os.urandom(30000)
def native(d, n, thr):
    paths=(C.c_char_p*n)(*[os.fsencode(os.path.join(d,f"n{thr}_{i}.b")) for i in range(n)])
    data=(C.c_char_p*n)(*[blob]*n); sizes=(C.c_int64*n)(*[len(blob)]*n)
    t0=time.perf_counter(); _lib.check(L.gpcc_write_files(paths,data,sizes,n,thr)); return time.perf_counter()-t0
def one(j):
    with open(j[0],'wb') as f: f.write(j[1])
def py(d,n,thr):
    jobs=[(os.path.join(d,f"p{thr}_{i}.b"),blob) for i in range(n)]
    t0=time.perf_counter()
    if thr==1:
        for j in jobs: one(j)
    else:
        with ThreadPoolExecutor(thr) as ex: list(ex.map(one,jobs))
    return time.perf_counter()-t0
for base in (None, "/dev/shm"):
    with tempfile.TemporaryDirectory(dir=base) as d:
        print("dir", d)
        for thr in (1,2,4,8,16):
            print(" native thr",thr, round(native(d,2338,thr)*1e3,1),"ms", " python thr",thr, round(py(d,2338,thr)*1e3,1),"ms")
