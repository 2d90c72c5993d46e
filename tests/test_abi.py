"""CPU checks of the drop-in boundary: libquickprefill.so loads without a GPU and exports every symbol that
include/quickprefill.h declares; the ctypes table binds exactly that set; no product module imports the oracle."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    src = open(os.path.join(ROOT, "include", "quickprefill.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(qp_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from quickvideo_amd import native
    if not os.path.exists(native.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    lib = native.load_library()
    names = header_functions()
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in quickprefill.h but not exported"
    assert sorted(native.SIGNATURES) == names, "ctypes table and header disagree"
    assert b"gfx950" in lib.qp_version()


def test_missing_library_fails_loudly():
    from quickvideo_amd import native
    with pytest.raises(native.QuickPrefillUnavailable):
        native.load_library(os.path.join(ROOT, "does_not_exist.so"))


def test_no_gpu_means_no_engine():
    """Without a GPU the product refuses to construct its operators (there is no CPU fallback)."""
    import torch
    from quickvideo_amd import native
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.QuickPrefillUnavailable):
        native.QuickPrefillOps()


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "quickvideo_amd")
    offenders = []
    for dp, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                if re.search(r"^\s*(from|import)\s+oracle\b", txt, flags=re.M) or "qp_oracle" in txt:
                    offenders.append(os.path.join(dp, f))
    for f in ("lvu",):
        d = os.path.join(ROOT, f)
        if os.path.isdir(d):
            for dp, _, files in os.walk(d):
                for ff in files:
                    if ff.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dp, ff)).read(), flags=re.M):
                        offenders.append(os.path.join(dp, ff))
    assert not offenders, offenders


@pytest.mark.parametrize("src_name,kernel,min_kernels", [("quickvideo_amd/csrc/qp_attn_s6.hip", "attn_fwd_kernel_s6", 4),
                                                         ("tools/experiments/qp_attn_s7.hip", "attn_fwd_kernel_s7", 2)])
def test_pipelined_attention_kernel_does_not_spill(src_name, kernel, min_kernels):
    """attn_fwd_kernel_s6 / _s7 issue LDS reads by hand and wait for them with counted s_waitcnt statements that carry no register
    operands; a compiler spill of a fragment register between the read and its wait would store a value that has not landed
    yet (and s7 owns AGPRs hipcc must not use as spill space).  The kernels must therefore compile without vector spills or
    scratch — checked from hipcc's resource remarks."""
    import shutil, subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    src = os.path.join(ROOT, src_name)
    r = subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S", "-o", os.devnull,
                        "-I" + os.path.join(ROOT, "quickvideo_amd", "csrc"), "-Rpass-analysis=kernel-resource-usage", src], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    blocks = re.split(r"remark: Function Name: ", r.stderr)[1:]
    seen = 0
    for b in blocks:
        if kernel not in b.splitlines()[0]:
            continue
        seen += 1
        assert int(re.search(r"VGPRs Spill: (\d+)", b).group(1)) == 0, b[:400]
        assert int(re.search(r"ScratchSize \[bytes/lane\]: (\d+)", b).group(1)) == 0, b[:400]
        assert int(re.search(r"VGPRs: (\d+)", b).group(1)) <= 256
    assert seen >= min_kernels


C_CALLER = os.path.join(ROOT, "tests", "c", "abi_prune_attn.c")


def build_c_caller(out, source=None):
    """gcc, plain C99: the boundary is a C ABI (no C++ types, no torch types), and a C program links against the library."""
    import subprocess
    from quickvideo_amd import native
    lib_dir = os.path.dirname(native.LIB_PATH)
    cmd = ["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"), source or C_CALLER, "-o", out,
           "-L" + lib_dir, "-lquickprefill", "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + lib_dir, "-Wl,-rpath,/opt/rocm/lib"]
    p = subprocess.run(cmd, capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]


def test_header_is_plain_c_and_library_has_no_torch_dependency(tmp_path):
    import subprocess
    from quickvideo_amd import native
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", os.path.join(ROOT, "include", "quickprefill.h")],
                       capture_output=True, text=True)
    assert p.returncode == 0, p.stderr
    needed = subprocess.run(["readelf", "-d", native.LIB_PATH], capture_output=True, text=True).stdout
    libs = re.findall(r"NEEDED.*\[(.*?)\]", needed)
    assert libs and not [l for l in libs if re.search(r"torch|c10|python", l)], libs
    build_c_caller(str(tmp_path / "abi_c"))                     # compiles and links here; runs on the GPU box (tests/test_gpu_ops.py)
    build_c_caller(str(tmp_path / "abi_seg"), os.path.join(ROOT, "tests", "c", "abi_segment.c"))


def test_hbm_bound_kernels_use_no_scratch(tmp_path):
    """The byte-moving kernels of the path (prune / select / gather, RoPE + append, RMSNorm / SwiGLU glue) must keep their rows in
    registers: scratch memory is HBM-backed, so a kernel that parks staged rows there moves every byte twice — exactly what the
    first in-place prune did until rocprofv3's FETCH_SIZE / WRITE_SIZE showed 2.0x the algorithmic traffic.  Cross-compiles the
    three files to gfx950 assembly (no GPU needed) and reads each kernel's private segment size and spill counts."""
    import re
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("hipcc not available")
    csrc = os.path.join(ROOT, "quickvideo_amd", "csrc")
    seen = 0
    for name in ("qp_prune", "qp_rope", "qp_elementwise"):
        out = tmp_path / f"{name}.s"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", str(out),
                        os.path.join(csrc, f"{name}.hip")], check=True, capture_output=True, timeout=600)
        text = out.read_text()
        for m in re.finditer(r"\.name:\s+(\S+)\n(?:.*\n)*?\s+\.private_segment_fixed_size:\s+(\d+)\n(?:.*\n)*?\s+\.vgpr_spill_count:\s+(\d+)", text):
            kernel, private, spills = m.group(1), int(m.group(2)), int(m.group(3))
            assert private == 0 and spills == 0, f"{name}: {kernel} uses {private} B of scratch per lane ({spills} spilled VGPRs)"
            seen += 1
    assert seen >= 15


def test_segment_structs_have_the_same_layout_in_c_and_in_the_ctypes_mirror(tmp_path):
    """struct qp_layer / struct qp_segment cross the boundary BY LAYOUT (qp_prefill_segment): a field added on one side only would shift
    every pointer behind it.  gcc prints sizeof / offsetof of every field from include/quickprefill.h; the ctypes mirror in
    quickvideo_amd/native.py must agree field by field."""
    import subprocess
    from quickvideo_amd.native import QpLayer, QpSegment
    src = tmp_path / "layout.c"
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "quickprefill.h"', 'int main(void) {']
    for cname, cls in (("qp_layer", QpLayer), ("qp_segment", QpSegment)):
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for f, _t in cls._fields_:
            lines.append(f'  printf("{cname}.{f} %zu\\n", offsetof({cname}, {f}));')
    lines += ['  return 0;', '}']
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    p = subprocess.run(["gcc", "-std=c99", "-Wall", "-Werror", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", "-I" + os.path.join(ROOT, "include"),
                        str(src), "-o", str(exe)], capture_output=True, text=True)
    assert p.returncode == 0, p.stderr[-2000:]
    got = dict(l.split() for l in subprocess.run([str(exe)], capture_output=True, text=True).stdout.splitlines())
    import ctypes
    for cname, cls in (("qp_layer", QpLayer), ("qp_segment", QpSegment)):
        assert int(got[cname]) == ctypes.sizeof(cls), (cname, got[cname], ctypes.sizeof(cls))
        for f, _t in cls._fields_:
            assert int(got[f"{cname}.{f}"]) == getattr(cls, f).offset, (cname, f)
