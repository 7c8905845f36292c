import sys; sys.path.insert(0,'/root/repo')
from __graft_entry__ import load_package
pkg=load_package()
for nrec in (64, 256):
    ms=pkg._lib.write_probe(4096,1024,nrec,3)
    gb=4096*nrec*2*1024*8/1e9
    print(nrec, "ms",ms,"TB/s", gb/ms)
