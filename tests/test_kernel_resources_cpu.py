"""CPU: no hot kernel may spill to scratch memory.

hipcc's `-Rpass-analysis=kernel-resource-usage` remarks are the compiler's own account of every kernel of a translation unit (registers,
spills, scratch bytes per lane).  A spill inside a 432-MFMA chunk body is a scratch load / store on the critical path of kernels that
run at the chip's power limit; round 3 shipped `conv3x3r_kernel<4, true>` with 68 bytes per lane and
`attn_bwd_kv_kernel<4, 4>` with 184 without any test noticing.  This test compiles the hot sources exactly as pdae_amd/build.py does and
asserts `ScratchSize == 0` for every kernel in them (the instantiations the F128 / C64 plans launch are a subset), except the ones listed
in ALLOWED with the reason.  `tools/kernel_resources.py` prints the same table by hand."""
import os
import re
import subprocess
import tempfile
from concurrent.futures import ThreadPoolExecutor

import pytest

from pdae_amd import build as B

HOT = ["conv3x3y.hip", "conv3x3r.hip", "conv3x3p.hip", "conv3x3w.hip", "conv3x3v.hip", "conv1x1.hip", "attention.hip", "convedge.hip", "norm.hip"]
# kernel-name regex -> why a scratch allocation is tolerated there
ALLOWED = {
    r"conv3x3y_kernelILi2ELb1E": "three-product bf16 split (PDAE_CONV_MATH=bf16x3) with fused GroupNorm input: 24 bytes, not on the default path",
    r"conv3x3p_kernelILi3ELi16ELb0ELb1E": "bf16x6 fallback arithmetic (PDAE_CONV_MATH=bf16x6 / after a saturation event), 16-row tiles: 8 bytes, not on the default path",
}


_ASM = {}      # src -> device assembly text of the same compilation (for the hazard scan below)


def resource_table(src):
    """[{name, vgpr, agpr, sgpr_spill, vgpr_spill, scratch, occupancy}] of every kernel in csrc/<src>."""
    with tempfile.TemporaryDirectory() as td:
        cmd = [B.HIPCC] + B.FLAGS + ["-Rpass-analysis=kernel-resource-usage", "--save-temps=obj", "-c", os.path.join(B.CSRC, src), "-o", os.path.join(td, "o.o")]
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=td)
        for fn in os.listdir(td):
            if fn.endswith(".s") and "amdgcn" in fn:
                _ASM[src] = open(os.path.join(td, fn)).read()
    assert r.returncode == 0, r.stderr[-2000:]
    rows, cur = [], None
    for line in r.stderr.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        t = re.sub(r"^\S+:\d+:\d+:\s*", "", m.group(1).strip())      # (with --save-temps the source location follows the word "remark:")
        if t.startswith("Function Name:"):
            cur = {"name": t.split(":", 1)[1].strip()}
            rows.append(cur)
        elif cur is not None and ":" in t:
            k, v = t.split(":", 1)
            cur[k.strip()] = v.strip()
    out = []
    for c in rows:
        out.append(dict(name=c["name"], vgpr=int(c.get("VGPRs", 0)), agpr=int(c.get("AGPRs", 0)), sgpr_spill=int(c.get("SGPRs Spill", 0)),
                        vgpr_spill=int(c.get("VGPRs Spill", 0)), scratch=int(c.get("ScratchSize [bytes/lane]", 0)),
                        occupancy=int(c.get("Occupancy [waves/SIMD]", 0)), lds=int(c.get("LDS Size [bytes/block]", 0))))
    return out


@pytest.mark.timeout(900)
def test_hot_kernels_use_no_scratch_memory():
    srcs = [s for s in HOT if os.path.exists(os.path.join(B.CSRC, s))]
    assert len(srcs) >= 7
    with ThreadPoolExecutor(max_workers=len(srcs)) as ex:
        tables = dict(zip(srcs, ex.map(resource_table, srcs)))
    bad, n = [], 0
    for src, rows in tables.items():
        assert rows, f"no kernels reported for {src}"
        for k in rows:
            n += 1
            if k["scratch"] > 0 and not any(re.search(p, k["name"]) for p in ALLOWED):
                bad.append((src, k["name"], k["scratch"], k["vgpr_spill"]))
    assert n > 60
    assert not bad, f"kernels with scratch memory (source, kernel, bytes per lane, spilled VGPRs): {bad}"
    # the instantiations the F128 training / sampling plans launch most (profiles/r03_kernel_stats.txt) must be in the table at all
    names = " ".join(k["name"] for rows in tables.values() for k in rows)
    for must in ("conv3x3y_kernelILi4ELb1E", "conv3x3y_kernelILi4ELb0E", "conv3x3r_kernelILi4ELb1E", "conv3x3r_kernelILi4ELb0E", "attn_bwd_kv_kernelILi4ELi4E", "conv3x3w_kernelILi4E", "conv3x3v_kernelILi4ELb0E", "conv3x3v_kernelILi4ELb1E", "conv1x1_kernelILi4E"):
        assert must in names, must


from pdae_amd.isa_hazard import lds_store_hazard_sites, output_store_hazard_sites      # noqa: E402  (product-side: the build runs them too)


@pytest.mark.timeout(900)
def test_no_output_store_data_hazard_sites_in_conv3x3y():
    """The kernel in which the corruption was observed (and every instantiation of it) must not contain the pattern; the other hot sources are scanned
    and reported: conv3x3r (the PDAE_W1 = 0 / bf16x6 fallback) has such sites in its drain, has never been seen to fail (43 bit-identity cases against
    conv3x3p) and is listed in DESIGN.md section 6 as an open item."""
    if "conv3x3y.hip" not in _ASM:
        resource_table("conv3x3y.hip")
    sites = output_store_hazard_sites(_ASM["conv3x3y.hip"])
    assert not sites, sites[:4]
    demo = "_Zk:\n\tbuffer_store_dwordx4 v[162:165], v82, s[28:31], s11 offen\n\tv_pk_add_f32 v[92:93], v[24:25], v[80:81]\n\tv_pk_add_f32 v[162:163], v[22:23], v[78:79]\n"
    assert len(output_store_hazard_sites(demo)) == 1
    for src in ("conv3x3r.hip", "conv3x3p.hip"):
        if src in _ASM:
            print(f"[store-data pattern] {src}: {len(output_store_hazard_sites(_ASM[src]))} sites (informational)")


@pytest.mark.timeout(900)
def test_no_lds_store_source_hazard_sites_in_hot_kernels():
    srcs = [s for s in HOT if os.path.exists(os.path.join(B.CSRC, s))]
    missing = [s for s in srcs if s not in _ASM]
    if missing:
        with ThreadPoolExecutor(max_workers=len(missing)) as ex:
            list(ex.map(resource_table, missing))
    bad = {s: lds_store_hazard_sites(_ASM[s])[:3] for s in srcs if lds_store_hazard_sites(_ASM[s])}
    assert not bad, f"multi-dword LDS stores whose last data register is overwritten before the store has drained: {bad}"
    # the scan does see the pattern (the shape that failed on the GPU)
    demo = "_Zk:\n\tds_write2_b32 v40, v41, v42 offset1:36\n\tv_sub_f32_e32 v41, v43, v44\n\tv_accvgpr_read_b32 v42, a98\n"
    assert len(lds_store_hazard_sites(demo)) == 1
    demo5 = "_Zk:\n\tds_write2_b32 v157, v174, v176 offset0:44 offset1:224\n\tv_add_u32_e32 v176, 0x400, v83\n\tds_write2_b32 v176, v175, v177 offset0:32 offset1:68\n"
    assert len(lds_store_hazard_sites(demo5)) == 1


def mfma_loop_waits(asm, kernel_re):
    """{kernel: [vmcnt values of the s_waitcnt instructions inside its innermost loops that contain MFMAs]} for the kernels matching kernel_re.
    Vector-memory results return in order: a wait for an L2-resident weight fragment that was requested AFTER an HBM prefetch drains the prefetch
    (conv1x1) -- a small vmcnt inside an MFMA loop that also prefetches is the signature."""
    out, kern, body, label, last_label = {}, None, None, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1) if re.search(kernel_re, m.group(1)) else None
            body = None
            continue
        if kern is None:
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            last_label = m.group(1)
        if "Inner Loop Header" in line:                # (on the label's line, or on a comment line of its own under a "Parent Loop" line)
            body, label = [], last_label
            continue
        t = line.strip().split(";")[0].strip()
        if body is None or not t:
            continue
        body.append(t)
        if t.startswith("s_cbranch") and t.split()[-1] == label:          # the loop's back edge
            if any(b.startswith("v_mfma") for b in body):
                out.setdefault(kern, []).extend(int(re.search(r"vmcnt\((\d+)\)", b).group(1)) for b in body if b.startswith("s_waitcnt") and "vmcnt" in b)
            body = None
    return out


@pytest.mark.timeout(900)
def test_conv1x1_stage_loop_does_not_drain_its_prefetch():
    """The buffer-load instantiations of conv1x1_kernel: inside the stage loop every fragment wait leaves the eight prefetch loads of the next stage
    (and the six younger refills) in flight -- vmcnt(14) --, the hand-over waits for the prefetch with the eight refills behind it in flight, and
    nothing in the loop waits for an empty queue.  (Round 5: a two-slot weight ring made every step wait with vmcnt(2), i.e. for the whole prefetch.)"""
    if "conv1x1.hip" not in _ASM:
        resource_table("conv1x1.hip")
    waits = mfma_loop_waits(_ASM["conv1x1.hip"], r"conv1x1_kernelILi\dELb0E")
    assert len(waits) == 4, sorted(waits)
    for k, w in waits.items():
        planes = {"1": 1, "2": 2, "4": 2, "3": 3}[re.search(r"ILi(\d)E", k).group(1)]
        # a step's wait: the 8 prefetch loads + the refills of the three other ring slots (the compiler may issue a slot's refill before or after the wait)
        assert w and min(w) >= 3 * planes and sum(v >= 8 + 3 * planes - 1 for v in w) >= 3, (k, w)


def mfma_loop_bodies(asm, kernel_re):
    """{kernel: [instruction lists of its innermost loops that contain MFMAs]} (same walk as mfma_loop_waits)."""
    out, kern, body, label, last_label = {}, None, None, None, None
    for line in asm.split("\n"):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern = m.group(1) if re.search(kernel_re, m.group(1)) else None
            body = None
            continue
        if kern is None:
            continue
        m = re.match(r"^(\.LBB\w+):", line)
        if m:
            last_label = m.group(1)
        if "Inner Loop Header" in line:
            body, label = [], last_label
            continue
        t = line.strip().split(";")[0].strip()
        if body is None or not t:
            continue
        body.append(t)
        if t.startswith("s_cbranch") and t.split()[-1] == label:
            if any(b.startswith("v_mfma") for b in body):
                out.setdefault(kern, []).append(body)
            body = None
    return out


@pytest.mark.timeout(900)
def test_conv3x3v_matrix_waves_issue_nothing_but_mfmas_and_lds_reads():
    """conv3x3v.hip (round 6): the design property that makes the producer / consumer split worth having, pinned in the compiled kernel.  The tile loop
    of a matrix wave (one per tap half: two MFMA loops per instantiation) holds exactly the tile's MFMAs -- 216 for the two-plane formats (36 tap
    steps x 2 output-channel tiles x 3 products), 72 for one plane -- and their 208 / 104 transposing LDS reads, ONE barrier, and no vector-memory
    instruction, no vmcnt wait and at most a handful of VALU instructions (the three buffer-base adds): everything else lives in the staging waves'
    loop, which in turn holds no MFMA."""
    if "conv3x3v.hip" not in _ASM:
        resource_table("conv3x3v.hip")
    loops = mfma_loop_bodies(_ASM["conv3x3v.hip"], r"conv3x3v_kernel")
    assert len(loops) == 6, sorted(loops)                                  # NS in {1, 2, 4} x GN in {0, 1}
    for kern, bodies in loops.items():
        ns = int(re.search(r"kernelILi(\d)E", kern).group(1))
        n_mfma, n_read = (72, 104) if ns == 1 else (216, 208)
        assert len(bodies) == 2, (kern, len(bodies))                       # tap halves 0 and 1
        for b in bodies:
            assert sum(1 for t in b if t.startswith("v_mfma")) == n_mfma, kern
            assert sum(1 for t in b if t.startswith("ds_read")) == n_read, kern
            assert sum(1 for t in b if t.startswith("s_barrier")) == 1, kern
            assert not [t for t in b if t.startswith(("buffer_", "global_", "flat_", "scratch_", "ds_write"))], kern
            assert not [t for t in b if t.startswith("s_waitcnt") and "vmcnt" in t], kern
            valu = [t for t in b if t.startswith("v_") and not t.startswith("v_mfma")]
            assert len(valu) <= 8, (kern, valu)
