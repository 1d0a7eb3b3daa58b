"""Registers, scratch, LDS and code size of every kernel in the SHIPPED library, read from the gfx950 code object
inside multi-purpose-mpc_amd/csrc/libmpmpc.so (llvm-readelf --notes of the unbundled .hip_fatbin).

    python profiles/kernel_resources.py            # prints the table
    python profiles/kernel_resources.py r3         # ... and writes profiles/r3/kernel_resources.txt

tests/test_abi.py imports kernel_table() and fails when a batch-path solve kernel has scratch
(private_segment_fixed_size != 0) or exceeds its register / LDS budget.
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
SO = os.path.join(ROOT, "multi-purpose-mpc_amd", "csrc", "libmpmpc.so")


def _run(*cmd):
    return subprocess.run(cmd, check=True, capture_output=True, text=True).stdout


def _demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names) + "\n", capture_output=True, text=True, check=True).stdout
    return [re.sub(r"^void ", "", l.split("(")[0]) for l in out.splitlines()]


def code_object(so=SO, workdir=None):
    """Path of the gfx950 ELF carved out of the library's .hip_fatbin section."""
    d = workdir or tempfile.mkdtemp(prefix="mpmpc_co_")
    fat, co = os.path.join(d, "fatbin"), os.path.join(d, "gfx950.co")
    _run(os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", so, fat)
    _run(os.path.join(LLVM, "clang-offload-bundler"), "--type=o", "--unbundle", "--input=" + fat, "--output=" + co,
         "--targets=hipv4-amdgcn-amd-amdhsa--gfx950")
    return co


def kernel_table(so=SO):
    """[{name, vgpr, agpr, sgpr, scratch, lds, code_bytes}] for every kernel of the code object."""
    co = code_object(so)
    notes = _run(os.path.join(LLVM, "llvm-readelf"), "--notes", co)
    syms = _run(os.path.join(LLVM, "llvm-readelf"), "-s", "-W", co)
    size = {}
    for l in syms.splitlines():
        f = l.split()
        if len(f) >= 8 and f[3] == "FUNC":
            size[f[7]] = int(f[2])
    rows, cur = [], None
    key = {".name": "mangled", ".vgpr_count": "vgpr", ".agpr_count": "agpr", ".sgpr_count": "sgpr",
           ".private_segment_fixed_size": "scratch", ".group_segment_fixed_size": "lds"}
    for l in notes.splitlines():
        m = re.match(r"\s*(-\s+)?(\.[a-z_]+):\s+(.*)$", l)
        if not m:
            continue
        if m.group(1) and l.startswith("  - "):          # a new entry of amdhsa.kernels
            cur = {}
            rows.append(cur)
        if cur is not None and m.group(2) in key:
            v = m.group(3).strip().strip("'")
            cur[key[m.group(2)]] = v if m.group(2) == ".name" else int(v)
    rows = [r for r in rows if "mangled" in r and "vgpr" in r]
    for r, n in zip(rows, _demangle([r["mangled"] for r in rows])):
        r["name"] = n
        r["code_bytes"] = size.get(r["mangled"], 0)
        r.setdefault("agpr", 0)
    return sorted(rows, key=lambda r: r["name"])


def render(rows, version=""):
    head = "%-62s %5s %5s %5s %8s %7s %9s" % ("kernel", "vgpr", "agpr", "sgpr", "scratch", "lds", "code")
    lines = ["# llvm-readelf --notes of the gfx950 code object in libmpmpc.so  %s" % version,
             "# vgpr + agpr <= 512 / waves per SIMD; scratch = private_segment_fixed_size (bytes per lane); lds, code in bytes",
             head]
    for r in rows:
        lines.append("%-62s %5d %5d %5d %8d %7d %9d" % (r["name"], r["vgpr"], r["agpr"], r["sgpr"], r["scratch"], r["lds"], r["code_bytes"]))
    return "\n".join(lines) + "\n"


if __name__ == "__main__":
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g
    rows = kernel_table()
    txt = render(rows, "(src %s)" % g.source_hash())
    print(txt, end="")
    if len(sys.argv) > 1:
        out = os.path.join(ROOT, "profiles", sys.argv[1])
        os.makedirs(out, exist_ok=True)
        open(os.path.join(out, "kernel_resources.txt"), "w").write(txt)
