#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of libamid_hip.so as hipcc reports it (-Rpass-analysis=kernel-resource-usage), one line
per kernel:  python profiles/tools/kernel_resources.py > profiles/r04_kernel_resources.txt   (no GPU needed: one cross-compile per source)."""
import glob, os, re, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "amid_amd", "csrc")
FILT = "/usr/bin/c++filt"


def file_flags(path):
    """The per-file flags of the product build (csrc/Makefile FLAGS_<stem> = ...): the listing must describe the kernels that ship."""
    stem = os.path.splitext(os.path.basename(path))[0]
    for ln in open(os.path.join(SRC, "Makefile")):
        m = re.match(r"FLAGS_" + re.escape(stem) + r"\s*=\s*(.*)", ln)
        if m:
            return m.group(1).split()
    return []


def one(path):
    with tempfile.NamedTemporaryFile(suffix=".o") as tmp:
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I" + os.path.join(ROOT, "include"),
                            *file_flags(path), "-Rpass-analysis=kernel-resource-usage", "-c", path, "-o", tmp.name], capture_output=True, text=True, cwd=SRC)
    rows, cur = [], None
    for ln in r.stderr.splitlines():
        m = re.search(r"remark:\s+(Function Name|TotalSGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", ln)
        if not m:
            continue
        k, v = m.group(1), m.group(2)
        if k == "Function Name":
            cur = {"name": v}
            rows.append(cur)
        elif cur is not None:
            cur[k.split(" ")[0]] = v
    return os.path.basename(path), rows


files = sorted(glob.glob(os.path.join(SRC, "*.hip")))
with ThreadPoolExecutor(4) as ex:
    res = list(ex.map(one, files))
names = [r["name"] for _, rows in res for r in rows]
dem = subprocess.run([FILT], input="\n".join(names), capture_output=True, text=True).stdout.splitlines() if names else []
dem = dict(zip(names, dem))
print("# file | kernel | VGPRs AGPRs SGPRs | scratch B/lane | static LDS B | waves/SIMD     (hipcc -O3 --offload-arch=gfx950)")
n_scratch = 0
for f, rows in res:
    for r in rows:
        d = re.sub(r"\(.*", "", dem.get(r["name"], r["name"]))
        d = d.replace("void ", "").replace("amid::", "")
        sc = int(r.get("ScratchSize", 0))
        n_scratch += sc > 0
        print(f"{f:24s} {d[:70]:70s} v{r.get('VGPRs', '?'):>3s} a{r.get('AGPRs', '?'):>3s} s{r.get('TotalSGPRs', '?'):>3s}  scratch {sc:>3d}  lds {r.get('LDS', '?'):>6s}  occ {r.get('Occupancy', '?')}")
print(f"# kernels with scratch: {n_scratch}")
