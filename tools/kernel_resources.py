#!/usr/bin/env python
"""Per-kernel resources of the built library, read from the code objects inside ``libtvl_hip.so`` (no GPU needed):
VGPRs / AGPRs / SGPRs, LDS bytes, and the private segment ("scratch": register spills and stack) of every kernel.

    python tools/kernel_resources.py [--scratch-only] [--check] [lib.so]

``--check`` is the build gate (csrc/Makefile): a kernel of the product library may own a private segment only if it is on ``ALLOWED`` below,
and in a DMA-ring kernel (which counts its LDS-DMA requests on ``vmcnt`` by hand: scratch traffic rides the same counter) no scratch
instruction may lie between the kernel's first ``global_load_lds`` and the barrier behind its last one -- spills belong to the epilogue, after the
ring has drained.  Exit code 1 otherwise.

The fat binary sections hold one clang offload bundle per translation unit; the gfx950 entry of each is an ELF whose
``NT_AMDGPU_METADATA`` note lists the kernels (``llvm-readelf --notes``)."""
from __future__ import annotations

import re
import struct
import subprocess
import sys
import tempfile
from pathlib import Path

READELF = "/opt/rocm/lib/llvm/bin/llvm-readelf"
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
# kernels that may spill: the 256-row tile of the two-piece ring GEMM sits at 254-256 VGPRs and parks 4-12 accumulator registers for its
# epilogue (measured faster than the 192-row tile for dz and the large convs).  Everything else must fit its registers.
ALLOWED = re.compile(r"^gemm_h2m_kernel<256, -?\d+, false, (true|false), false>$")
RING = re.compile(r"^gemm_(h2m|tp3)_kernel<")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib: Path):
    data = lib.read_bytes()
    pos = 0
    while True:
        i = data.find(MAGIC, pos)
        if i < 0:
            return
        n = struct.unpack_from("<Q", data, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", data, off)
            triple = data[off + 24: off + 24 + tl].decode()
            off += 24 + tl
            if "gfx" in triple and size:
                yield triple, data[i + o: i + o + size]
        pos = i + 24


def scratch_placement(co_path: str, symbols: set[str]) -> dict[str, str]:
    """symbol -> complaint for every ring kernel of ``symbols`` with a scratch instruction in front of its last s_barrier."""
    txt = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", co_path], capture_output=True, text=True).stdout
    bad, cur, insns = {}, None, []

    def close():
        if cur in symbols and insns:
            dma = [i for i, t in enumerate(insns) if t.startswith("global_load_lds")]
            if not dma:
                return
            # the ring: from its first LDS-DMA request to the barrier that follows the last one (the step's B_t after the final vmcnt(0))
            ring_end = next((i for i in range(dma[-1], len(insns)) if insns[i].startswith("s_barrier")), dma[-1])
            inside = [i for i, t in enumerate(insns) if t.startswith("scratch_") and dma[0] <= i <= ring_end]
            if inside:
                bad[cur] = f"{len(inside)} scratch instruction(s) inside the DMA ring (instructions {dma[0]}..{ring_end} of {len(insns)}; first at {inside[0]})"

    for line in txt.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            close()
            cur, insns = m.group(1), []
        elif cur and line.startswith(("\t", "  ")) and line.strip():
            insns.append(line.strip().split("//")[0].strip())
    close()
    return bad


def kernels(lib: Path, check_placement: bool = False):
    out = []
    for triple, blob in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(blob)
            f.flush()
            txt = subprocess.run([READELF, "--notes", f.name], capture_output=True, text=True).stdout
            ring_syms = set()
            if check_placement:
                blocks = re.findall(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", txt, re.S)
                for blk in blocks:
                    nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
                    if int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", blk).group(1)) and "gemm_" in nm:
                        ring_syms.add(nm)
                placement = scratch_placement(f.name, ring_syms) if ring_syms else {}
            else:
                placement = {}
        for m in re.finditer(r"- \.agpr_count:.*?(?=\n  - \.agpr_count:|\namdhsa\.target|\Z)", txt, re.S):
            blk = m.group(0)
            g = lambda k: (re.search(rf"\.{k}:\s+(\S+)", blk) or [None, "?"])[1]  # noqa: E731
            name = g("name")
            try:
                name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
            except OSError:
                pass
            out.append({"placement": placement.get(g("name")), "name": re.sub(r"^void \(anonymous namespace\)::", "", name).split("(")[0], "vgpr": g("vgpr_count"), "agpr": g("agpr_count"), "sgpr": g("sgpr_count"),
                        "lds": g("group_segment_fixed_size"), "scratch": g("private_segment_fixed_size"), "spill_v": g("vgpr_spill_count"), "spill_s": g("sgpr_spill_count")})
    return out


def main():
    if not (Path(READELF).exists() and Path(OBJDUMP).exists()):   # (a toolchain without the LLVM binutils: the gate cannot run; say so, do not fail the build)
        print(f"kernel_resources: {READELF} / {OBJDUMP} not found -- resource check skipped")
        return
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    lib = Path(args[0]) if args else Path(__file__).resolve().parents[1] / "tunevlseg_amd" / "csrc" / "libtvl_hip.so"
    check = "--check" in sys.argv
    ks = kernels(lib, check_placement=check)
    only = "--scratch-only" in sys.argv or check
    if check:
        bad = [k for k in ks if k["scratch"] not in ("0", "?") and not ALLOWED.match(k["name"])]
        misplaced = [k for k in ks if k["placement"]]
        for k in bad:
            print(f"kernel_resources: {k['name']} owns a private segment of {k['scratch']} B/lane (spills v{k['spill_v']} s{k['spill_s']}) and is not on the allow-list")
        for k in misplaced:
            print(f"kernel_resources: {k['name']}: {k['placement']} -- scratch traffic inside a hand-counted vmcnt ring")
        if bad or misplaced:
            sys.exit(1)
    print(f"{len(ks)} kernels in {lib.name}; {sum(1 for k in ks if k['scratch'] not in ('0', '?'))} with a private segment")
    for k in sorted(ks, key=lambda k: (-int(k["scratch"]) if k["scratch"].isdigit() else 0, k["name"])):
        if only and k["scratch"] == "0":
            continue
        print(f"{k['scratch']:>6} B scratch  vgpr {k['vgpr']:>3} agpr {k['agpr']:>3} sgpr {k['sgpr']:>3} lds {k['lds']:>6}  spills v{k['spill_v']} s{k['spill_s']}  {k['name']}")


if __name__ == "__main__":
    main()
