"""The C-ABI library loads and exports every symbol include/tvl_hip.h declares (no compute calls, CPU-only)."""
import re
from pathlib import Path

import torch  # noqa: F401  (the library binds to the HIP runtime torch loaded)

ROOT = Path(__file__).resolve().parents[1]


def declared_symbols():
    hdr = (ROOT / "include" / "tvl_hip.h").read_text()
    return sorted(set(re.findall(r"\b(tvl_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol():
    from tunevlseg_amd import hip

    lib = hip.load()
    assert lib.tvl_abi_version() == 6
    decl = declared_symbols()
    assert len(decl) >= 30
    missing = [s for s in decl if not hasattr(lib, s)]
    assert not missing, missing


def test_python_binding_covers_header_one_to_one():
    from tunevlseg_amd import hip

    decl = set(declared_symbols())
    assert decl == set(hip.EXPORTS)


def test_struct_layouts_match_header(tmp_path):
    """ctypes mirrors of tvlGemmArgs / tvlAttn*Args agree with what a C compiler lays out from the header."""
    import ctypes as C
    import shutil
    import subprocess

    import pytest

    from tunevlseg_amd import hip

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    probe = tmp_path / "probe.c"
    fields = {"tvlGemmArgs": ["layout", "A", "ldb", "C", "bias", "residual", "act", "pre_out", "dact_aux", "dact", "alpha", "a_map", "c_map"],
              "tvlGemmTp3Args": ["M", "A", "a_rows", "B", "C", "ldc", "C_tp3", "bias", "residual", "ldr", "act", "pre_out", "dact_aux", "ld_aux", "dact", "alpha",
                                 "tile_m", "variant", "workspace", "workspace_bytes", "aux_blocked", "a_scale_one"],
              "tvlAttnFwdArgs": ["q", "q_bs", "q_ts", "o", "ldo", "lse", "key_mask", "B", "causal", "scale", "Tk"],
              "tvlAttnBwdArgs": ["q", "v_ts", "o", "d_o", "ldo", "lse", "delta", "dq", "dq_bs", "dv_ts", "key_mask", "B", "scale", "Tk"]}
    lines = ['#include <stdio.h>', '#include <stddef.h>', f'#include "{ROOT / "include" / "tvl_hip.h"}"', "int main(void){"]
    for st, fs in fields.items():
        lines.append(f'printf("{st} %zu\\n", sizeof({st}));')
        lines += [f'printf("{st}.{f} %zu\\n", offsetof({st}, {f}));' for f in fs]
    lines.append("return 0;}")
    probe.write_text("\n".join(lines))
    exe = tmp_path / "probe"
    subprocess.run(["gcc", str(probe), "-o", str(exe)], check=True)
    out = dict(l.split() for l in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    mirror = {"tvlGemmArgs": hip.GemmArgs, "tvlGemmTp3Args": hip.GemmTp3Args, "tvlAttnFwdArgs": hip.AttnFwdArgs, "tvlAttnBwdArgs": hip.AttnBwdArgs}
    for st, fs in fields.items():
        assert int(out[st]) == C.sizeof(mirror[st]), st
        for f in fs:
            assert int(out[f"{st}.{f}"]) == getattr(mirror[st], f).offset, (st, f)


def test_product_path_fails_loudly_without_a_gpu():
    """No CPU fallback: a CPU tensor reaching a HIP op raises (the oracle is never used by the product path)."""
    import pytest

    from tunevlseg_amd import hip

    if torch.cuda.is_available():
        pytest.skip("CPU-only check")
    x = torch.zeros(4, 8)
    with pytest.raises(RuntimeError):
        hip.layernorm_fwd(x, torch.ones(8), torch.zeros(8), 1e-5)


def test_product_never_imports_oracle():
    pkg = ROOT / "tunevlseg_amd"
    for f in pkg.rglob("*.py"):
        assert "oracle" not in f.read_text().replace("CPU oracle", ""), f


def test_no_kernel_outside_the_allow_list_owns_scratch():
    """tools/kernel_resources.py --check on the built library (the gate csrc/Makefile runs at link time): a private segment only in the 256-row
    tile of the ring GEMM, and there no scratch instruction inside the hand-counted DMA ring."""
    import subprocess
    import sys

    from tunevlseg_amd import hip

    r = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_resources.py"), "--check", str(hip.LIB_PATH)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "with a private segment" in r.stdout
