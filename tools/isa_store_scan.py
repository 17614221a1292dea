"""Scan of the compiled kernels for serialised stores: a global store that is preceded by `s_waitcnt vmcnt(0)` while an
earlier store of the same kernel is still outstanding waits for that store's whole round trip.  hipcc produces the pattern
when an HBM operand is first consumed inside a predicated block (block GEMM epilogues).

usage: python tools/isa_store_scan.py            # compiles every csrc/*.hip with -save-temps into /tmp/vc_isa
"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = "/tmp/vc_isa"
os.makedirs(OUT, exist_ok=True)
procs = []
for src in sorted(glob.glob(os.path.join(ROOT, "voicecraft_amd", "csrc", "*.hip"))):
    stem = os.path.splitext(os.path.basename(src))[0]
    procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-c", src,
                                   "-o", f"{OUT}/{stem}.o", "-save-temps=obj"], cwd=os.path.dirname(src),
                                  stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL))
for p in procs:
    p.wait()
hits = 0
for f in sorted(glob.glob(f"{OUT}/*-hip-amdgcn-amd-amdhsa-gfx950.s")):
    s = open(f).read()
    ks = list(re.finditer(r"^(_Z\S+):\s*; @", s, re.M))
    for i, m in enumerate(ks):
        body = s[m.end(): ks[i + 1].start() if i + 1 < len(ks) else len(s)]
        seq = []
        for line in body.splitlines():
            t = line.strip()
            if t.startswith(("global_store", "buffer_store")):
                seq.append("S")
            elif t.startswith("s_waitcnt vmcnt(0)"):
                seq.append("W")
            elif t.startswith(("global_load", "buffer_load")):
                seq.append("L")
        st = "".join(seq)
        n = len(re.findall(r"S(?=W+S)", st))
        if n >= 2:
            hits += 1
            name = m.group(1)
            for filt in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
                try:
                    name = subprocess.run([filt, m.group(1)], capture_output=True, text=True).stdout.strip() or name
                    break
                except OSError:
                    pass
            print(f"{os.path.basename(f).split('-hip-')[0]:10s} {n:3d} of {st.count('S'):3d} stores wait for the previous one: {name[:110]}")
print(f"{hits} kernel(s) with serialised store chains")
