"""Static checks on the gfx950 ISA hipcc emits for kernels.hip (CPU only, cross-compile)."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def device_asm(tmp_path_factory):
    from vectorlite_amd import build as vbuild
    out = tmp_path_factory.mktemp("isa") / "kernels.s"
    cmd = [vbuild.hipcc(), f"--offload-arch={vbuild.ARCH}"] + vbuild.COMMON + [
        "--cuda-device-only", "-S", os.path.join(vbuild.CSRC, "kernels.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    return out.read_text()


@pytest.fixture(scope="module")
def mfma_asm(tmp_path_factory):
    """mfma_scan.hip's device assembly, compiled once for the tests below (two minutes of hipcc)."""
    from vectorlite_amd import build as vbuild
    out = tmp_path_factory.mktemp("isa_mfma") / "mfma.s"
    cmd = [vbuild.hipcc(), f"--offload-arch={vbuild.ARCH}"] + vbuild.COMMON + [
        "--cuda-device-only", "-S", os.path.join(vbuild.CSRC, "mfma_scan.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True)
    return out.read_text()


def test_no_unencodable_64bit_scalar_literal(device_asm):
    # ROCm 7.2 hipcc can emit `s_mov_b64 s[a:b], <64-bit literal>` for gfx950, which the hardware
    # encoding truncates to 32 bits (found on MI355X: a -inf f64 threshold became +0.0).
    bad = re.findall(r"s_mov_b64 s\[\d+:\d+\], 0x[0-9a-f]{9,}", device_asm)
    assert bad == []


def _kernel_body(asm, mangled_fragment):
    m = re.search(r"^(_Z\S*%s\S*):[^\n]*\n(.*?)s_endpgm" % re.escape(mangled_fragment), asm, flags=re.S | re.M)
    assert m, mangled_fragment
    return m.group(2)


def test_exact_kernels_do_not_contract_multiply_add(device_asm):
    # the reference's f64 loops are separate multiply and add (src/lib.rs:430-434, 479-482, 568-571):
    # the accumulation loops must use v_mul_f64 / v_add_f64; v_fma_f64 may only appear in the
    # correctly rounded sqrt/division expansions after the loop.
    for frag in ("12k_exact_scanILi0E", "12k_exact_scanILi1E", "12k_exact_scanILi3E"):
        body = _kernel_body(device_asm, frag)
        assert "v_mul_f64" in body and "v_add_f64" in body
        last_mul = body.rfind("v_mul_f64")
        first_fma = body.find("v_fma_f64")
        assert first_fma == -1 or "v_rsq_f64" in body[:first_fma] or "v_rcp_f64" in body[:first_fma], frag
        assert last_mul != -1


def test_scan_kernel_streams_with_16_byte_nontemporal_loads(device_asm):
    body = _kernel_body(device_asm, "6k_scanILi0ELi8ELi12ELi1E")  # default shape for dim = 384
    nt = re.findall(r"global_load_dwordx4 .* nt", body)
    assert len(nt) == 12  # 12 float4 per lane (one row group of 8 rows) in flight
    assert "scratch_" not in body and "buffer_store" not in body  # no spills
    meta = re.search(r"\.name:\s+_ZN2vl12_GLOBAL__N_16k_scanILi0ELi8ELi12ELi1E.*?\.vgpr_count:\s+(\d+)", device_asm, re.S)
    if meta:
        assert int(meta.group(1)) <= 128


def test_mfma_scan_default_shapes_do_not_spill(mfma_asm):
    """The batched bf16 MFMA kernels keep 96-192 registers of query fragments per lane; a scheduling
    change once made the sampling pass spill 5000 registers (2.4x slower).  The launch shapes used by
    default (8 waves x 1 tile for dim <= 512, 4 waves x 1 tile for dim 768) must stay spill-free."""
    asm = mfma_asm
    seen = 0
    for block in asm.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        m = re.search(r"k_mfma_scanILi(\d+)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E", name)
        if not m:
            continue
        ksteps, mode, metric, nw, qt = map(int, m.groups())
        default = (ksteps < 48 and (nw, qt) == (8, 1)) or (ksteps >= 48 and (nw, qt) == (4, 1))
        if not default:
            continue
        seen += 1
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1))
        assert scratch <= 128, (name, scratch)
        assert "v_mfma_f32_16x16x32_bf16" in asm
    assert seen >= 24


def test_row_stationary_kernel_shapes_in_the_default_build(mfma_asm):
    """k_mfma_rows (K4r), the shipped batch filter: the default build holds ONLY the two-waves-per-SIMD shape (RBN 2, NW 8) --
    the one-wave-per-SIMD shape measured 8-28 % slower in round 4 and compiles in with -DRS_WIDE_SHAPES alone -- and its
    pass-1 instantiations stay within the spills they are known to have (Euclidean at stride 512: 25 registers -- 33 before
    its per-query-block thresholds moved to LDS in round 4, which freed stride 384 of its 10; everything else none)."""
    asm = mfma_asm
    seen = 0
    for block in asm.split("- .agpr_count:")[1:]:
        name = re.search(r"\.name:\s+(\S+)", block).group(1)
        m = re.search(r"k_mfma_rowsILi(\d+)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELb([01])E", name)
        if not m:
            continue
        ksteps, mode, metric, rbn, nw, stream = map(int, m.groups())
        assert (rbn, nw) == (2, 8), name
        assert not (stream and mode == 0), name     # nontemporal row loads: single-chunk launches of pass 1 only
        seen += 1
        scratch = int(re.search(r"\.private_segment_fixed_size:\s+(\d+)", block).group(1))
        vgpr = int(re.search(r"\.vgpr_count:\s+(\d+)", block).group(1))
        assert vgpr <= 256, (name, vgpr)            # two waves per SIMD
        allowed = 128 if (metric == 1 and ksteps == 32 and mode == 1) else 0
        assert scratch <= allowed, (name, scratch)
    assert seen == 45                                # 5 strides x 3 metrics x (sampling, pass 1, pass 1 with streaming loads)
