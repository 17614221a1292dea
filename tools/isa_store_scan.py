"""Scan of the compiled kernels for serialised stores: a global store that is preceded by `s_waitcnt vmcnt(0)` while an
earlier store of the same kernel is still outstanding waits for that store's whole round trip.  hipcc produces the pattern
when an HBM operand is first consumed inside a predicated block (block GEMM epilogues).

usage: python tools/isa_store_scan.py            # compiles every csrc/*.hip to gfx950 ISA (device only) into /tmp/vc_isa
(tests/test_isa.py runs the same scan, plus scratch / instruction checks, on every CPU test run)
"""
import glob, os, re, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def compile_isa(outdir):
    """gfx950 ISA of every translation unit (device side only, ~35 s in parallel) -> {stem: text}"""
    os.makedirs(outdir, exist_ok=True)
    procs = []
    for src in sorted(glob.glob(os.path.join(ROOT, "voicecraft_amd", "csrc", "*.hip"))):
        stem = os.path.splitext(os.path.basename(src))[0]
        dst = os.path.join(outdir, stem + ".s")
        procs.append((stem, dst, subprocess.Popen([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "--cuda-device-only",
                                                   "-S", src, "-o", dst], cwd=os.path.dirname(src),
                                                  stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, text=True)))
    out = {}
    for stem, dst, p in procs:
        _, err = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {stem}: {err[-2000:]}")
        out[stem] = open(dst).read()
    return out


def kernels(text):
    """{mangled name: (body, scratch bytes, vgprs)} of one ISA file"""
    heads = list(re.finditer(r"^(_Z\S+):\s*; @", text, re.M))
    res = {}
    for i, m in enumerate(heads):
        body = text[m.end(): heads[i + 1].start() if i + 1 < len(heads) else len(text)]
        meta = re.search(r"\.amdhsa_kernel " + re.escape(m.group(1)) + r"\n(.*?)\.end_amdhsa_kernel", text, re.S)
        if not meta:
            continue                        # a device function, not a kernel
        g = lambda k: int(re.search(k + r"\s+(\d+)", meta.group(1)).group(1))
        res[m.group(1)] = (body, g(r"\.amdhsa_private_segment_fixed_size"), g(r"\.amdhsa_next_free_vgpr"))
    return res


def store_chains(body):
    """(stores that wait for the previous store, stores) of one kernel body"""
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
    return len(re.findall(r"S(?=W+S)", st)), st.count("S")


def demangle(name):
    for filt in ("/opt/rocm/lib/llvm/bin/llvm-cxxfilt", "c++filt"):
        try:
            return subprocess.run([filt, name], capture_output=True, text=True).stdout.strip() or name
        except OSError:
            pass
    return name


if __name__ == "__main__":
    hits = 0
    for stem, text in compile_isa("/tmp/vc_isa").items():
        for name, (body, scratch, vgpr) in kernels(text).items():
            n, total = store_chains(body)
            if n >= 2:
                hits += 1
                print(f"{stem:10s} {n:3d} of {total:3d} stores wait for the previous one: {demangle(name)[:110]}")
    print(f"{hits} kernel(s) with serialised store chains")
