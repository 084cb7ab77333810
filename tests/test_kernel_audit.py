"""Static audit of the fused MLP kernels' hand-counted LDS waits (tools/audit_lds_waits.py).

The fp16-pair kernel reads LDS from inline asm and wait with hand-counted `s_waitcnt lgkmcnt(N)`; hipcc neither counts
those reads nor knows their destinations are in flight. The audit walks the generated assembly of every input mode
of the kernel and fails if any instruction touches a register before the wait that retires its read - which covers both
a wait whose count is too large and a compiler move / spill of a destination register. CPU only: hipcc cross-compiles
the device code to assembly (no GPU, the in-tree library is not touched)."""
import importlib.util
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "nerf-projects_amd")


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


KERNELS = {
    # the three input modes of the inference kernel and the training forward kernel (rays; STORE = 1 row-major, 2 blocked by
    # 32 points with nt stores)
    "mlp_kernel_h2.hip": (("kernelILi0ELi0E", "kernelILi1ELi0E", "kernelILi2ELi0E", "kernelILi2ELi1E", "kernelILi2ELi2E"), 1000, 400),
    # the training backward-data kernel on the same machinery, row-major and blocked
    # (<blocked, view-dependent>: the chain without view directions is 12 chunks shorter)
    "mlp_bwd_kernel_h2.hip": (("nerf_mlp_bwd_h2_kernelILb0ELb1E", "nerf_mlp_bwd_h2_kernelILb1ELb1E", "nerf_mlp_bwd_h2_kernelILb0ELb0E",
                               "nerf_mlp_bwd_h2_kernelILb1ELb0E"), 400, 200),
}


@pytest.mark.parametrize("src", sorted(KERNELS))
def test_no_register_touched_before_its_lds_wait(src, tmp_path):
    build = _load(os.path.join(PKG, "build.py"), "nerf_build_for_audit")
    audit = _load(os.path.join(ROOT, "tools", "audit_lds_waits.py"), "audit_lds_waits")
    out = tmp_path / (src + ".s")
    cmd = [build.hipcc()] + build.FLAGS + build.EXTRA.get(src, build.VGPR_FORM) + \
        ["-I", os.path.join(ROOT, "include"), "-I", build.CSRC, "--cuda-device-only", "-S",
         os.path.join(build.CSRC, src), "-o", str(out)]
    subprocess.run(cmd, check=True, cwd=tmp_path)
    names, min_ops, min_waits = KERNELS[src]
    for inst in names:
        findings, n_ops, n_waits = audit.audit(str(out), inst)
        assert n_ops > min_ops and n_waits > min_waits, (inst, n_ops, n_waits)   # the kernel was found and parsed
        assert not findings, (inst, findings[:5])
        # ... and no inline-asm store / atomic reads a scalar base a vector instruction (e.g. an SGPR-spill reload) has just
        # written: the five wait states hipcc would insert for its own instructions (this was a GPU memory fault once)
        hazards = audit.audit_sgpr_hazards(str(out), inst)
        assert not hazards, (inst, hazards[:5])
    # no scratch: a spilled register is reloaded through the vector-memory counter, which the weight ring's waits own
    text = open(out).read()
    for inst in names:
        body = text[text.index(inst):]
        body = body[:body.index("s_endpgm")]
        assert "scratch_" not in body, inst


def test_weight_gradient_kernel_prefetched_through_lds(tmp_path):
    """grad_batch_pair_dma_kernel (train_dw_kernel.hip) keeps two steps of operands in flight as LDS-DMA loads and reads them
    back from inline asm behind hand-counted waits: the same audit (no register of a read back touched before its wait, no
    scalar hazard), no scratch (a reload would go through the vector-memory counter the prefetch owns), and the only counted
    vector-memory waits of the step loop are the ones the source states - 16 loads of a step, 17 with the rider row, 0."""
    import re
    build = _load(os.path.join(PKG, "build.py"), "nerf_build_for_audit")
    audit = _load(os.path.join(ROOT, "tools", "audit_lds_waits.py"), "audit_lds_waits")
    src = "train_dw_kernel.hip"
    out = tmp_path / (src + ".s")
    cmd = [build.hipcc()] + build.FLAGS + build.EXTRA.get(src, build.VGPR_FORM) + \
        ["-I", os.path.join(ROOT, "include"), "-I", build.CSRC, "--cuda-device-only", "-S",
         os.path.join(build.CSRC, src), "-o", str(out)]
    subprocess.run(cmd, check=True, cwd=tmp_path)
    text = open(out).read()
    for inst in ("grad_batch_pair_dma_kernelILb0E", "grad_batch_pair_dma_kernelILb1E"):      # row-major / blocked operands
        findings, n_ops, n_waits = audit.audit(str(out), inst)
        assert n_ops > 40 and n_waits > 15, (inst, n_ops, n_waits)
        assert not findings, (inst, findings[:5])
        assert not audit.audit_sgpr_hazards(str(out), inst), inst
        body = text[text.index(inst):]
        body = body[:body.index("s_endpgm")]
        assert "scratch_" not in body, inst
        assert body.count("global_load_lds_dwordx4") >= 32 and body.count("global_load_lds_dword ") >= 1, inst
        # (the predicated tail - register loads, behind the drained ring - brings a small count of hipcc's own)
        waits = set(re.findall(r"s_waitcnt vmcnt\((\d+)\)", body))
        assert {"0", "16", "17"} <= waits and all(int(w) <= 2 for w in waits - {"16", "17"}), (inst, waits)
    # the gamma columns' kernel: eight dY images + four KiB of X per step, twelve loads a step
    inst = "grad_batch_narrow_pair_kernel"
    findings, n_ops, n_waits = audit.audit(str(out), inst)
    assert n_ops > 20 and n_waits > 1, (inst, n_ops, n_waits)
    assert not findings, (inst, findings[:5])
    assert not audit.audit_sgpr_hazards(str(out), inst), inst
    body = text[text.index(inst):]
    body = body[:body.index("s_endpgm")]
    assert "scratch_" not in body, inst
    assert body.count("global_load_lds_dwordx4") >= 12, inst
    # (the tail's predicated register loads bring hipcc's own small counts; the loop's wait is the one the source states)
    waits = re.findall(r"s_waitcnt vmcnt\((\d+)\)", body)
    assert waits.count("12") == 1 and "0" in waits and all(int(w) <= 12 for w in waits), (inst, waits)
