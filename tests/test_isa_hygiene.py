"""Properties of the generated gfx950 code that the performance of the hot kernels rests on, checked on the ISA hipcc emits
(cross-compiled here, no GPU): each of them was once silently lost and cost a measurable fraction of the kernel.

  * conv_wino_kernel: no register spills in the shipped (two-part) form; no FLAT loads anywhere in the kernel -- a flat request
    possibly pending in the tile loop turns every counted wait of the loop into s_waitcnt vmcnt(0); the chunk body holds exactly
    72 (two fp16 parts) / 144 (three bf16 parts) matrix instructions and starts behind a COUNTED wait;
  * every 16-byte inline-assembly store is followed by two wait states (gfx950: a store of more than 8 bytes must not be followed
    by a vector write of its data registers within two wait states; hipcc adds them for its own stores only).
"""
import os
import re
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "motif_amd", "csrc")


@pytest.fixture(scope="module")
def wino_isa(tmp_path_factory):
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not on PATH")
    out = tmp_path_factory.mktemp("isa") / "conv_wino.s"
    cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-value", "-Wno-pass-failed",
           "-I", CSRC, "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S", os.path.join(CSRC, "conv_wino.hip"), "-o", str(out)]
    subprocess.run(cmd, check=True, capture_output=True, timeout=900)
    text = out.read_text()
    kernels = {}
    for m in re.finditer(r"^(_Z16conv_wino_kernelILi(\d)ELb(\d)ELb(\d)EEv8ConvArgsii):", text, re.M):
        end = text.index(".Lfunc_end", m.end())
        kernels[(int(m.group(2)), bool(int(m.group(3))), bool(int(m.group(4))))] = text[m.end():end].splitlines()
    meta = {}
    for m in re.finditer(r"\.name:\s+(_Z16conv_wino_kernelILi(\d)ELb(\d)ELb(\d)EEv8ConvArgsii)\s", text):
        blk = text[max(0, m.start() - 1500):m.end() + 1500]
        sp = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk)
        meta[(int(m.group(2)), bool(int(m.group(3))), bool(int(m.group(4))))] = int(sp.group(1)) if sp else None
    m = re.search(r"^_Z22conv_wino_chain_kernel8ConvArgsii9ChainArgs:", text, re.M)
    kernels["chain"] = text[m.end():text.index(".Lfunc_end", m.end())].splitlines()
    return kernels, meta


def _instr(lines):
    for ln in lines:
        t = ln.strip()
        if t and not t.startswith((";", ".", "/")) and not t.endswith(":"):
            yield t


def test_conv_wino_kernels_exist_without_flat_loads_and_without_spills_in_the_shipped_form(wino_isa):
    kernels, meta = wino_isa
    # (parts, multi-problem, transposed accumulators): the two-part form in both accumulator layouts, the three-part form row-major
    assert set(kernels) == {(2, False, False), (2, True, False), (2, False, True), (2, True, True), (3, False, False), (3, True, False), "chain"}
    for key, lines in kernels.items():
        ops = [t.split()[0] for t in _instr(lines)]
        assert not [o for o in ops if o.startswith("flat_")], "conv_wino_kernel %s has FLAT memory instructions" % (key,)
    for multi in (False, True):
        for tr in (False, True):
            assert meta[(2, multi, tr)] == 0, "the two-part kernel must not spill vector registers (%s)" % meta
    # the transposed form's epilogue stays out of LDS: no more LDS instructions than the row-major form's chunk body + staging alone
    lds = {key: sum(1 for t in _instr(lines) if t.startswith("ds_")) for key, lines in kernels.items() if key != "chain"}
    assert lds[(2, False, True)] < lds[(2, False, False)] // 2, lds


@pytest.mark.parametrize("parts, mfmas, tr", [(2, 72, False), (2, 72, True), (3, 144, False)])
def test_conv_wino_chunk_body_is_one_run_of_matrix_instructions_behind_a_counted_wait(wino_isa, parts, mfmas, tr):
    kernels, _ = wino_isa
    lines = kernels[(parts, False, tr)]
    idx = [i for i, ln in enumerate(lines) if "v_mfma_f32_32x32x16" in ln]
    runs, start, prev, n = [], idx[0], idx[0], 1
    for i in idx[1:]:
        if i - prev > 80:
            runs.append((start, prev, n)); start, n = i, 0
        n += 1; prev = i
    runs.append((start, prev, n))
    body = max(runs, key=lambda r: r[2])
    assert body[2] == mfmas, runs
    opcode = "v_mfma_f32_32x32x16_f16" if parts == 2 else "v_mfma_f32_32x32x16_bf16"
    assert all(opcode in lines[i] for i in idx if body[0] <= i <= body[1])
    waits = [t for t in _instr(lines[body[0] - 60:body[0]]) if t.startswith("s_waitcnt") and "vmcnt" in t]
    assert waits, "the chunk's first fragments wait for their weights"
    assert "vmcnt(0)" not in waits[-1], "every chunk would begin by waiting for ALL requests in flight: %s" % waits


def test_wide_inline_assembly_stores_are_followed_by_two_wait_states(wino_isa):
    kernels, _ = wino_isa
    for key, lines in kernels.items():
        ins = list(_instr(lines))
        stores = [i for i, t in enumerate(ins) if t.startswith("global_store_dwordx4")]
        assert stores, key
        for i in stores:
            nxt = ins[i + 1]
            assert nxt.startswith("s_nop") and int(nxt.split()[1]) >= 1, "conv_wino_kernel %s: %s / %s" % (key, ins[i], nxt)


def test_conv_wino_chain_kernel_signals_through_scalar_mailboxes_and_moves_activations_device_coherently(wino_isa):
    """The chain kernel (motif_conv2d_chain_fwd): (i) its tickets and completion words are scalar atomics whose results land in s100 / s101
    a chunk or a tile before they are used -- the compiler must never allocate those two registers (it stops at s99 on gfx950; were that to
    change, a result could arrive in a register holding something else); (ii) activations, residuals and outputs move with sc1 (the XCDs' L2s
    are not coherent inside a kernel), weights stay cached; (iii) the chunk body is the same single run of 72 matrix instructions."""
    kernels, _ = wino_isa
    lines = kernels["chain"]
    in_asm, mailbox_outside, atomics = False, [], []
    for ln in lines:
        t = ln.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
        elif t.startswith(";;#ASMEND"):
            in_asm = False
        elif re.search(r"\bs10[01]\b|s\[100:101\]|s\[9\d:10[01]\]", t) and not t.startswith(";"):
            if not in_asm:
                mailbox_outside.append(t)
            if t.startswith("s_atomic_add"):
                atomics.append(t)
    assert not mailbox_outside, "compiler-generated code touches a mailbox register: %s" % mailbox_outside[:3]
    assert len(atomics) >= 3 and all(t.endswith("glc") and re.match(r"s_atomic_add s10[01],", t) for t in atomics), atomics
    ins = list(_instr(lines))
    stores = [t for t in ins if t.startswith("global_store_dwordx4")]
    assert stores and all(t.endswith("sc1") for t in stores), [t for t in stores if not t.endswith("sc1")][:3]
    loads = [t for t in ins if t.startswith("buffer_load_dwordx4")]
    coherent = [t for t in loads if t.endswith("sc1")]
    assert len(coherent) >= 30 and len(loads) - len(coherent) >= 12, (len(loads), len(coherent))      # row pieces + residual quads | weight fragments
    assert not [t for t in ins if t.startswith("flat_")]
    assert sum(1 for t in ins if t.startswith("scratch_load")) <= 4, "spill reloads in the chain kernel wait for every store in flight"
    idx = [i for i, ln in enumerate(lines) if "v_mfma_f32_32x32x16_f16" in ln]
    runs, start, prev, n = [], idx[0], idx[0], 1
    for i in idx[1:]:
        if i - prev > 80:
            runs.append(n); n = 0
        n += 1; prev = i
    runs.append(n)
    assert max(runs) == 72, runs


@pytest.fixture(scope="module")
def other_isa(tmp_path_factory):
    """Device assembly of the other two-part kernels (compiled side by side)."""
    if not shutil.which("hipcc"):
        pytest.skip("hipcc not on PATH")
    from concurrent.futures import ThreadPoolExecutor
    d = tmp_path_factory.mktemp("isa2")

    def comp(name):
        out = d / (name + ".s")
        cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-munsafe-fp-atomics", "-Wno-unused-value", "-Wno-pass-failed",
               "-I", CSRC, "-I", os.path.join(ROOT, "include"), "--cuda-device-only", "-S", os.path.join(CSRC, name + ".hip"), "-o", str(out)]
        subprocess.run(cmd, check=True, capture_output=True, timeout=900)
        return out.read_text()
    with ThreadPoolExecutor(max_workers=3) as ex:
        texts = dict(zip(("conv_pw", "dcn", "siren_split"), ex.map(comp, ("conv_pw", "dcn", "siren_split"))))
    meta = {}
    for text in texts.values():
        for m in re.finditer(r"\.name:\s+(\S+)\s", text):
            blk = text[max(0, m.start() - 1500):m.end() + 1500]
            sp, vg = re.search(r"\.vgpr_spill_count:\s+(\d+)", blk), re.search(r"\.vgpr_count:\s+(\d+)", blk)
            if sp and vg:
                meta[m.group(1)] = (int(vg.group(1)), int(sp.group(1)))
    return meta


def test_the_shipped_two_part_kernels_do_not_spill(other_isa):
    """conv_pw_kernel (all cout-tile counts; its first version needed 242 registers and spilled), the window DCN and the three SIREN
    networks in their two-part forms: no vector-register spills, and the pointwise kernel within the 128 registers that let two
    8-wave workgroups share a CU."""
    want = ["_Z14conv_pw_kernelILi%dEEv8ConvArgsiiil" % t for t in (1, 2, 3, 4)] + ["_Z14dcn_win_kernelILi8ELi2EEv12DcnFusedArgs"] + \
           ["_Z18siren_split_kernelILi%dELi1ELi2EEv9SirenArgs" % m for m in (0, 1, 2, 3)]
    for k in want:
        assert k in other_isa, (k, sorted(other_isa)[:40])
        assert other_isa[k][1] == 0, (k, other_isa[k])
    for t in (1, 2, 3):
        assert other_isa["_Z14conv_pw_kernelILi%dEEv8ConvArgsiiil" % t][0] <= 128, other_isa


# ---------------------------------------------------------------------------------------------- packed fp32 forms (round 6)
# `v_pk_fma_f32 d, a, b, c op_sel:[0,1,0]` with b a vector-register pair (the LOW result takes b's HIGH register) gave wrong low halves in lanes
# 48..63 while an fp16 / bf16 MFMA kernel ran beside it on another stream (tools/pk_fma_beside_mfma.hip reproduces it without the
# library; profiles/r06_pk_fma_beside_mfma.txt); the same form on a SCALAR register pair, the plain form and the op_sel_hi forms never did.
# hipcc picks the form by itself, so the check is on what was BUILT -- every gfx950 code object inside libmotif_hip.so, disassembled -- and it is
# wider than the one form seen to fail: no packed fp32 instruction may have an op_sel bit on a vector-register source.
PK_F32 = re.compile(r"\b(v_pk_(?:fma|mul|add)_f32)\s+([^/;]*?)\s+op_sel:\[([01,]+)\]")


def pk_low_from_high_vector_register(line):
    m = PK_F32.search(line)
    if not m:
        return False
    sources = [o.strip() for o in m.group(2).split(",")][1:]
    return any(bit == "1" and i < len(sources) and sources[i].startswith("v") for i, bit in enumerate(m.group(3).split(",")))


def test_no_kernel_of_the_built_library_feeds_a_packed_fp32_low_result_from_the_high_register_of_a_vector_pair(tmp_path):
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    lib = os.path.join(ROOT, "motif_amd", "libmotif_hip.so")
    if not os.path.exists(objdump) or not os.path.exists(lib):
        pytest.skip("needs llvm-objdump and the built library")
    assert pk_low_from_high_vector_register("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel:[0,1,0]")
    assert pk_low_from_high_vector_register("v_pk_mul_f32 v[0:1], v[2:3], s[4:5] op_sel:[1,0] op_sel_hi:[0,1]")
    assert not pk_low_from_high_vector_register("v_pk_fma_f32 v[0:1], v[2:3], s[4:5], v[0:1] op_sel:[0,1,0]")         # the scalar pair: never seen to fail
    assert not pk_low_from_high_vector_register("v_pk_fma_f32 v[0:1], v[2:3], v[4:5], v[0:1] op_sel_hi:[1,0,1]")
    copy = tmp_path / "libmotif_hip.so"
    shutil.copy(lib, copy)
    subprocess.run([objdump, "--offloading", str(copy)], check=True, capture_output=True, timeout=300, cwd=tmp_path)
    objs = sorted(p for p in tmp_path.iterdir() if "amdgcn" in p.name)
    assert len(objs) >= 10, [p.name for p in objs]                      # one code object per kernel source
    bad, kernels = [], 0
    for obj in objs:
        text = subprocess.run([objdump, "-d", str(obj)], check=True, capture_output=True, timeout=600, text=True).stdout
        name = None
        for ln in text.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\w+)>:", ln)
            if m:
                name, kernels = m.group(1), kernels + 1
            elif pk_low_from_high_vector_register(ln):
                bad.append((name, ln.split("//")[0].strip()))
    assert kernels > 60
    assert not bad, "packed fp32 instructions with a low result from the high register of a vector pair (mark the kernel MOTIF_SCALAR_F32, common.h): %s" % bad[:6]
