"""Per-kernel resource use of one HIP source compiled for gfx950 (no GPU needed):
    python scripts/isa_stats.py touchnet_amd/csrc/gemm.hip [-D...]
prints VGPR / AGPR / SGPR counts, scratch bytes, spills, and the number of MFMA / LDS-DMA / barrier / `s_waitcnt vmcnt(0)`
instructions of every kernel (kernel-development aid: a fused epilogue must not push the main loop into scratch)."""
import os
import re
import subprocess
import sys
import tempfile

src = sys.argv[1]
extra = sys.argv[2:]
d = tempfile.mkdtemp(prefix="isa_")
base = os.path.basename(src)[:-4]
cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-Wno-unused-result", *extra, "-c",
       os.path.abspath(src), "-o", os.path.join(d, base + ".o"), "-save-temps=obj"]
subprocess.run(cmd, check=True, cwd=d)
asm = open(os.path.join(d, f"{base}-hip-amdgcn-amd-amdhsa-gfx950.s")).read()
meta = asm[asm.find("amdhsa.kernels"):]
bodies = {}
for m in re.finditer(r"^(_Z\w+):[^\n]*\n(.*?)\n\s+s_endpgm", asm, flags=re.S | re.M):
    bodies[m.group(1)] = m.group(2)
for e in meta.split("- .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", e).group(1)
    get = lambda k: (re.search(rf"\.{k}:\s+(\d+)", e) or [None, "-"])[1]
    dn = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    b = bodies.get(name, "")
    cnt = lambda pat: len(re.findall(pat, b))
    vm0 = cnt(r"vmcnt\(0\)")
    print(f"{dn[:120]}\n    vgpr {get('vgpr_count')} agpr {e.split()[1]} sgpr {get('sgpr_count')} scratch "
          f"{get('private_segment_fixed_size')} vspill {get('vgpr_spill_count')} sspill {get('sgpr_spill_count')} | "
          f"mfma {cnt(r'v_mfma')} lds-dma {cnt(r'buffer_load_dwordx4.* lds')} barrier {cnt(r's_barrier')} "
          f"vmcnt0 {vm0} lines {b.count(chr(10))}")
print("temps:", d)
