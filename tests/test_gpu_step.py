"""GPU parity of the whole distillation step (3 teachers + student, pseudo-labels, MTA + focal, backward,
Adam) against the golden vectors captured from the reference's ModelWithNMSLoss / ModelWithNMSKDListLoss."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.step import DistillEngine, StepConfig
from mm_distillnet_amd.synth import synth_inputs
from helpers import make_state, check_summary

DEV = "cuda"
BIAS = {"rgb": -2.0, "depth": -3.2, "thermal": -2.0}
MODS = {"rgb": (3, 21), "depth": (3, 22), "thermal": (1, 23)}


HEADER_W = "classifier.header.pointwise_conv.conv.weight"
# D4 at 768^2: the random-weight classifiers saturate (logit spread 3-10), so every teacher would emit ~5 000 labels and the
# ORACLE's focal loss would need a [110484 x 13227] IoU matrix (35 GB, minutes on the CPU); damping the header weights leaves
# 34 / 92 / 78 candidates per teacher (measured with the oracle)
D4_HEAD = {"rgb": (0.1, -2.0), "depth": (0.03, -2.0), "thermal": (0.03, -2.0)}


def teacher_states(coef, mods):
    out = {}
    for k, (cin, seed) in mods.items():
        if coef == 4:
            spec, st = make_state(coef, cin, seed, k, cls_bias=D4_HEAD[k][1])
            st[HEADER_W] = st[HEADER_W] * D4_HEAD[k][0]
        else:
            spec, st = make_state(coef, cin, seed, k, cls_bias=BIAS[k])
        out[k] = (spec, st)
    return out


def build(variant, S=256, precision="fp32", coef=2):
    mods = {"rgb": MODS["rgb"]} if variant == "rgb1" else MODS      # "rgb1" = BASELINE configs[1]: one RGB teacher
    teachers = teacher_states(coef, mods)
    spec_s, st_s = make_state(coef, 8, 24, "audio")
    cfg = StepConfig(image_size=S, kd_mode="list" if variant in ("list", "listaug") else "pairwise", augment=variant == "augmented",
                     precision=precision)
    eng = DistillEngine(spec_s, {k: v[0] for k, v in teachers.items()}, DEV, cfg)
    eng.load(st_s, {k: v[1] for k, v in teachers.items()})
    return eng, spec_s


def drop_scale_from(gold, spec):
    masks = {int(b): torch.from_numpy(m) for b, m in zip(gold["drop_blocks"], gold["drop_masks"])}
    skip = [b for b in spec.blocks if b.skip]
    return torch.stack([masks[b.idx] / (1.0 - b.drop_rate) for b in skip]).to(DEV)


def grad_checks(gold, grads, norm_tol, head_rtol, head_atol, fusion_atol=None):
    """fusion_atol: bound for the 2- / 3-element BiFPN fusion weights (`pN_wK`) where it differs: their gradient is a dot product over a
    whole map, and a max-pool tie on a small level moves it"""
    for k in gold.files:
        if k.startswith("gradnorm."):
            top = k[len("gradnorm."):]
            tot = sum(float(grads[n].double().pow(2).sum()) for n in grads
                      if (".".join(n.split(".")[:2]) if n.startswith("bifpn") else n.split(".")[0]) == top)
            assert abs(tot ** 0.5 - float(gold[k])) <= norm_tol * float(gold[k]) + 1e-9, (top, tot ** 0.5, float(gold[k]))
        if k.startswith("grad.") and k.endswith(".head"):
            name = k[5:-5]
            theta = fusion_atol is not None and name.split(".")[-1].startswith("p") and "_w" in name.split(".")[-1]
            check_summary(gold, "grad." + name, grads[name], head_rtol, fusion_atol if theta else head_atol)


def step_batch(variant, B, S):
    batch = {k: v.to(DEV) for k, v in synth_inputs(B, S, seed=31).items()}
    if variant == "listaug":      # ModelWithNMSKDListLossAugmented: `label` = RGB frames of other recordings, a 4th list entry
        batch["aug_rgb"] = synth_inputs(B, S, seed=57)["rgb"].to(DEV)
    return batch


@pytest.mark.parametrize("variant", ["pairwise", "list", "augmented", "rgb1", "listaug"])
def test_step_golden_reference_labels(golden_dir, variant):
    """Whole-step parity with the teacher-side integer noise removed: the teachers' pseudo-labels come from the reference run
    (the golden's per-teacher [n,6] rows) instead of the GPU teachers' decode + NMS, so no int() truncation of a last-bit-different
    box edge can move a label.  Everything downstream - cross-teacher merge NMS, anchor assignment, focal + smooth-L1, the MTA terms
    (GPU teacher forwards), the student's backward and Adam - is then held to fp32 tolerances: loss scalars 2e-4, gradients
    2e-3 of the tensor's largest value (fp32 summation order), Adam-updated weights 1e-5."""
    gold = np.load(os.path.join(golden_dir, f"step_d2_256_{variant}.npz"))
    S, B = 256, 2
    eng, spec = build(variant, S)
    nt = len(eng.teachers) + (1 if variant == "listaug" else 0)
    batch = step_batch(variant, B, S)
    ds = drop_scale_from(gold, spec)
    A = eng.student.anchors(S).shape[0]
    labels = eng.labels_from_rows([[gold[f"teacher{ti}_img{i}"] for i in range(B)] for ti in range(nt)], A)
    out = eng.step_body(batch, ds, teacher_labels=labels)
    torch.cuda.synchronize()
    eng.check_overflow()
    np.testing.assert_allclose(out["reg"].cpu().numpy(), gold["reg"], rtol=2e-4)
    np.testing.assert_allclose(out["cls"].cpu().numpy(), gold["cls"], rtol=2e-4)
    np.testing.assert_allclose(out["kd"].cpu().numpy(), gold["kd"].reshape(out["kd"].shape), rtol=1e-4, atol=1e-5)
    loss = 1.0 * (out["reg"].item() + out["cls"].item()) + 0.005 * out["kd"].sum().item()
    assert abs(loss - float(gold["loss"])) < 2e-4 * abs(float(gold["loss"]))
    grad_checks(gold, eng.student.ps.export_grads(), 2e-3, 2e-3, 2e-3)
    eng.optimizer_body()
    torch.cuda.synchronize()
    params = eng.student.ps.export_state()
    # (the first Adam step moves a weight by lr * g / (|g| + eps): sign-like, so an element whose gradient is of the order of eps = 1e-8
    # may land anywhere in [-lr, lr]; measured: one element of 64 off by 5e-6 = 0.05 lr with every gradient check above green)
    for k in gold.files:
        if k.startswith("adam.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(gold, "adam." + name, params[name], 1e-5, 1e-4)


@pytest.mark.parametrize("variant", ["pairwise", "list", "augmented", "rgb1", "listaug"])
def test_step_golden(golden_dir, variant):
    gold = np.load(os.path.join(golden_dir, f"step_d2_256_{variant}.npz"))
    S, B = 256, 2
    eng, spec = build(variant, S)
    nt = len(eng.teachers) + (1 if variant == "listaug" else 0)
    batch = step_batch(variant, B, S)
    ds = drop_scale_from(gold, spec)
    out = eng.step_body(batch, ds)
    torch.cuda.synchronize()
    eng.check_overflow()
    # teachers' pseudo-labels.  The kernels are bit-exact on identical inputs (test_gpu_losses.py); here the inputs are
    # the teachers' fp32 outputs computed on the GPU, which differ from the CPU run in the last bits, so a box edge that
    # sits on an integer boundary may truncate differently and a borderline candidate may flip: require >= 95 % of the
    # reference rows to have a counterpart within 1 px with the same label.
    tot = hit = 0
    exact = True          # every teacher's rows carry the reference run's integer box coordinates (as a set: the row order is the kernel's own)
    for ti in range(nt):
        for i in range(B):
            ref = gold[f"teacher{ti}_img{i}"]
            n = int(out["cnt_t"][ti][i].item())
            got = out["rows_t"][ti][i, :n].cpu().numpy()
            assert abs(n - ref.shape[0]) <= max(2, 0.05 * ref.shape[0]), (ti, i, n, ref.shape[0])
            exact = exact and n == ref.shape[0] and sorted(map(tuple, got[:, :4].tolist())) == sorted(map(tuple, ref[:, :4].tolist()))
            for r in ref:
                tot += 1
                if n and (np.abs(got[:, :4] - r[:4]).max(1) <= 1.0).any():
                    hit += 1
    assert hit >= 0.95 * tot, (hit, tot)
    print("losses", out["reg"].item(), gold["reg"], out["cls"].item(), gold["cls"], "labels exact:", exact)
    # One set of bounds, unconditionally.  The frozen teachers are bit-reproducible since round 5 (integer pool sums, fixed-order partials:
    # tests/test_gpu_net.py::test_frozen_nets_bit_reproducible), so this test's outcome no longer varies from run to run; the GPU teachers'
    # last bits still differ from the CPU reference's, which may move a borderline candidate in or out of the per-teacher row sets
    # (`exact` printed above) - the cross-teacher merge and the 2e-2 bounds below absorb that.  Parity on identical labels is what
    # test_step_golden_reference_labels holds to fp32 tolerances.
    grads = eng.student.ps.export_grads()
    loss = 1.0 * (out["reg"].item() + out["cls"].item()) + 0.005 * out["kd"].sum().item()
    lt = 2e-2
    np.testing.assert_allclose(out["reg"].cpu().numpy(), gold["reg"], rtol=lt)
    np.testing.assert_allclose(out["cls"].cpu().numpy(), gold["cls"], rtol=lt)
    np.testing.assert_allclose(out["kd"].cpu().numpy(), gold["kd"].reshape(out["kd"].shape), rtol=1e-4, atol=1e-5)
    assert abs(loss - float(gold["loss"])) < lt * abs(float(gold["loss"]))
    grad_checks(gold, grads, 2e-2, 3e-2, 2e-3)
    eng.optimizer_body()
    torch.cuda.synchronize()
    params = eng.student.ps.export_state()
    assert all(torch.isfinite(v).all() for v in params.values())
    for k in gold.files:
        if k.startswith("adam.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(gold, "adam." + name, params[name], 1e-4, 1e-4)
    # and the step is reproducible as far as the labels go: a second evaluation gives the same pseudo-label rows bit for bit
    out2 = eng.step_body(batch, ds)
    torch.cuda.synchronize()
    for ti in range(nt):
        assert torch.equal(out2["cnt_t"][ti], out["cnt_t"][ti])
        for i in range(B):
            n = int(out["cnt_t"][ti][i].item())
            assert torch.equal(out2["rows_t"][ti][i, :n], out["rows_t"][ti][i, :n]), (ti, i)


_D4_EMU = {}       # precision -> (gradient cos of the oracle's emulation to fp32, loss-shift scale) measured by the B = 2 case


@pytest.mark.parametrize("B", [2, 8])      # 8 = the per-GPU batch of the other configs (fp32 part only: ~7 GB of oracle tape per image)
def test_d4_768_step_vs_oracle(B):
    """BASELINE configs[4]'s architecture and input size through the WHOLE step: three frozen EfficientDet-D4 teachers + the
    8-channel D4 student at 768 x 768, pseudo-labels,
    MTA + focal losses, backward - in fp32 against oracle/step_ref (itself pinned at D4 by the reference-made goldens of round 5,
    tests/golden/step_d4_256_pairwise.npz and net_d4_*: test_step_d4_golden below, tests/test_oracle_golden.py::test_step_d4 - the
    reference's load_model hard-codes D2, its YetAnotherEfficientDet class does not), then the bf16 mixed-precision modes of the
    same step: at B = 2 bounded against the oracle's own emulation of each mode, at B = 8 (configs[4]'s per-GPU batch) against the
    fp32 HIP step with the B = 2 yardsticks, eagerly and through capture() + replay()."""
    from oracle import step_ref as ST
    from helpers import grad_state
    from mm_distillnet_amd.hostinfo import free_memory_gb
    S, coef = 768, 4
    if free_memory_gb() < 12 * B + 8:
        pytest.skip(f"D4 / 768² step test at B = {B} needs ~{12 * B + 8} GB of free host memory (have {free_memory_gb():.0f} GB): NOT RUN")
    eng, spec = build("pairwise", S, coef=coef)
    teachers = {k: v[1] for k, v in teacher_states(coef, MODS).items()}
    _, st = make_state(coef, 8, 24, "audio")
    so = grad_state(st)
    hb = synth_inputs(B, S, seed=33)
    batch = {k: v.to(DEV) for k, v in hb.items()}
    skip = [b for b in spec.blocks if b.skip]
    g = torch.Generator().manual_seed(3)
    masks = {b.idx: torch.floor((1.0 - b.drop_rate) + torch.rand(B, generator=g)) for b in skip}
    ds = torch.stack([masks[b.idx] / (1.0 - b.drop_rate) for b in skip]).to(DEV)
    ref = ST.distill_forward(so, teachers, hb, S, coef, masks)
    loss = ST.total_loss(ref)
    loss.backward()
    A = eng.student.anchors(S).shape[0]
    assert A == 110484
    nlab = [int(np.size(l) // 6) for t in ref["per_teacher"] for l in t]
    print("D4/768 oracle pseudo-labels per teacher:", nlab, "merged:", [int(np.size(l) // 5) for l in ref["labels"]])
    assert sum(nlab) > 0
    # (1) the GPU teachers' own labels: same count within 5 %, >= 95 % of the oracle's rows within 1 px
    out = eng.step_body(batch, ds)
    torch.cuda.synchronize()
    eng.check_overflow()
    tot = hit = 0
    for ti in range(3):
        for i in range(B):
            r = np.asarray(ref["per_teacher"][ti][i], dtype=np.float32).reshape(-1, 6)
            n = int(out["cnt_t"][ti][i].item())
            got = out["rows_t"][ti][i, :n].cpu().numpy()
            assert abs(n - r.shape[0]) <= max(2, 0.05 * r.shape[0]), (ti, i, n, r.shape[0])
            for row in r:
                tot += 1
                hit += int(n > 0 and (np.abs(got[:, :4] - row[:4]).max(1) <= 1.0).any())
    assert hit >= 0.95 * tot, (hit, tot)
    # (2) with the oracle's labels: losses 2e-4 (kd 1e-4), gradient direction / norm over all parameters
    labels_host = ref["per_teacher"]
    labels = eng.labels_from_rows(labels_host, A)
    out = eng.step_body(batch, ds, teacher_labels=labels)
    torch.cuda.synchronize()
    kd_ref = torch.stack(ref["kd"]).detach().numpy()
    np.testing.assert_allclose(out["reg"].cpu().numpy(), ref["reg"].detach().numpy(), rtol=2e-4)
    np.testing.assert_allclose(out["cls"].cpu().numpy(), ref["cls"].detach().numpy(), rtol=2e-4)
    np.testing.assert_allclose(out["kd"].cpu().numpy(), kd_ref.reshape(out["kd"].shape), rtol=1e-4, atol=1e-5)
    grads = eng.student.ps.export_grads()

    def compare(grads):
        dot = n1 = n2 = 0.0
        for k, v in so.items():
            if not v.requires_grad or v.grad is None:
                continue
            a, b_ = v.grad.double(), grads[k].double()
            dot += float((a * b_).sum()); n1 += float((a * a).sum()); n2 += float((b_ * b_).sum())
        return dot / (n1 ** 0.5 * n2 ** 0.5), (n2 / n1) ** 0.5

    cos, ratio = compare(grads)
    print("D4/768 fp32 step at B = %d, gradient: cos %.6f norm ratio %.5f" % (B, cos, ratio))
    assert cos > 0.9995 and abs(ratio - 1.0) < 5e-3, (cos, ratio)

    def cosine(ga, gb):
        d = a2 = b2 = 0.0
        for k, u in ga.items():
            if gb.get(k) is None:
                continue
            u, w = u.double(), gb[k].double()
            d += float((u * w).sum()); a2 += float((u * u).sum()); b2 += float((w * w).sum())
        return d / (a2 ** 0.5 * b2 ** 0.5)

    if B != 2:
        # (3') configs[4] at its OWN precision and per-GPU batch: bf16 at B = 8.  No oracle tape of the
        # emulation at this size (45 GB for fp32 alone): the yardsticks are the B = 2 run's emulation-derived numbers (_D4_EMU, filled by
        # the B = 2 case of this test; the constants are its round-3 measurements) and the fp32 HIP step of this batch.
        g32 = {k: v.clone() for k, v in grads.items()}
        loss32 = (out["reg"].item(), out["cls"].item(), out["kd"].cpu().numpy().copy())
        del eng, so, ref
        torch.cuda.empty_cache()
        for precision in ("bf16",):
            c_emu, shift = _D4_EMU.get(precision, {"bf16": (0.41, 0.11)}[precision])
            eng_b, _ = build("pairwise", S, precision=precision, coef=coef)
            ob = eng_b.step_body(batch, ds, teacher_labels=eng_b.labels_from_rows(labels_host, A))
            torch.cuda.synchronize()
            ghip = eng_b.student.ps.export_grads()
            assert all(torch.isfinite(v).all() for v in ghip.values())
            c_hip = cosine(ghip, g32)
            sh_cls, sh_reg = abs(ob["cls"].item() - loss32[1]) / abs(loss32[1]), abs(ob["reg"].item() - loss32[0]) / abs(loss32[0])
            print("D4/768 %s step at B = %d: gradient cos to the fp32 HIP gradient %.4f (B = 2 emulation vs fp32: %.4f), loss shift cls %.2e reg %.2e (B = 2 scale %.2e)" % (
                precision, B, c_hip, c_emu, sh_cls, sh_reg, shift))
            assert c_hip >= 0.75 * c_emu, (precision, c_hip, c_emu)
            assert sh_cls <= 2.0 * shift + 2e-2 and sh_reg <= 2.0 * shift + 2e-2, (precision, sh_cls, sh_reg, shift)
            np.testing.assert_allclose(ob["kd"].cpu().numpy(), loss32[2], rtol=0.1, atol=1e-3)
            # ... and through capture() + replay() with the GPU teachers' own labels: finite, and the replay reproduces the eager step
            oe = eng_b.step_body(batch, ds)
            torch.cuda.synchronize()
            le = (oe["reg"].item(), oe["cls"].item(), oe["kd"].cpu().numpy().copy(), oe["nbox"].cpu().tolist())
            be = [oe["boxes"][i, :le[3][i]].cpu().numpy() for i in range(B)]
            eng_b.capture(batch)
            orp = eng_b.replay(batch, ds)
            torch.cuda.synchronize()
            eng_b.check_overflow()
            nb = orp["nbox"].cpu().tolist()
            same = nb == le[3] and all(np.array_equal(orp["boxes"][i, :nb[i]].cpu().numpy(), be[i]) for i in range(B))
            # same pseudo-labels: only the f64 / fp32 atomics' summation order differs, which bf16 operand rounding amplifies to ~1e-3.
            # In these modes the TEACHERS run on the bf16 MFMA too: the same atomics noise flips operand roundings inside them, so two runs
            # of one teacher disagree on a few of its ~50 boxes per image (measured: reg 0.3412 vs 0.3302 with different label sets)
            # (a sanity bound in that case - 25 %: measured 3 - 9 % - the tight statement is the equal-labels one)
            # (round 5: the frozen nets are bit-reproducible, so the labels are equal in practice and what is left is the STUDENT's train-mode
            # atomics - BatchNorm sums, depthwise weight gradients - whose order differs between the eager and the captured schedule; the
            # bf16 operand rounding of the 32-block D4 student amplifies it: measured with equal labels reg 0.33532 vs 0.33741 = 6.2e-3 (bf16) and
            # 0.34093 vs 0.32519 = 4.8e-2 (bf16_hbm), cls 2e-2.  The yardstick is the mode's own distance from fp32, `shift` = the loss
            # shift of the oracle's emulation of the mode (0.11 at B = 2): two runs of the mode may differ by half of that)
            rt = (0.5 * shift + 1e-2) if same else 0.25
            print("D4/768 %s replay vs eager at B = %d: labels %s, reg %.6f / %.6f cls %.6f / %.6f" % (
                precision, B, "equal" if same else "differ (integer truncation)", orp["reg"].item(), le[0], orp["cls"].item(), le[1]))
            # (classification loss with other labels: these random-weight D4 students saturate, every anchor whose assignment changes moves the
            # focal sum by ~10 - measured 54 .. 112 over repeated eager bf16 steps of ONE state, tools/dev/diag_d4_labels.py - so only its order
            # of magnitude is checked then)
            assert abs(orp["reg"].item() - le[0]) <= rt * abs(le[0])
            assert abs(orp["cls"].item() - le[1]) <= rt * abs(le[1]) if same else 0.4 < orp["cls"].item() / le[1] < 2.5
            np.testing.assert_allclose(orp["kd"].cpu().numpy(), le[2], rtol=2e-2 if same else 0.5, atol=1e-4)
            assert torch.isfinite(eng_b.student.ps.grad).all() and torch.isfinite(eng_b.student.ps.flat).all()
            del eng_b
            torch.cuda.empty_cache()
        return
    # (3) bf16 mixed precision (configs[4]'s numerics): "bf16" = bf16 MFMA operands (the bf16 STORAGE mode "bf16_hbm" was deleted in round 6).
    # Same labels.  The yardstick is the ORACLE's own emulation of each mode (oracle/effdet_ref.py BF16_PW) run through the same step:
    # the rule itself moves the gradient by some angle from fp32; the HIP gradient must stay within twice that angle of both the fp32
    # gradient and the emulation's (two implementations of one rounding rule decorrelate: a last-bit difference upstream flips roundings
    # downstream), and the losses within twice the emulation's own loss shift.
    from oracle import effdet_ref as O
    loss32 = (out["reg"].item(), out["cls"].item(), out["kd"].cpu().numpy().copy())
    g32 = {k: v.grad.detach().clone() for k, v in so.items() if v.requires_grad and v.grad is not None}
    del eng

    emu = {}
    for precision in ("bf16",):
        se = grad_state(st)
        O.BF16_PW = True
        try:
            refe = ST.distill_forward(se, teachers, hb, S, coef, masks, per_teacher_labels=ref["per_teacher"])
            ST.total_loss(refe).backward()
        finally:
            O.BF16_PW = False
        emu[precision] = ({k: v.grad for k, v in se.items() if v.requires_grad and v.grad is not None},
                          max(abs(refe["cls"].item() - ref["cls"].item()) / abs(ref["cls"].item()),
                              abs(refe["reg"].item() - ref["reg"].item()) / abs(ref["reg"].item())))
    # The loss shift of a rounding mode on this random-weight D4 net is a noisy quantity: the emulation (deterministic) shifts by 11 % in one
    # mode and 4 % in the other, the HIP step (fp32 atomics upstream of the roundings) by 5 - 11 % from run to run of ONE mode.  The bound
    # therefore takes the larger of the two emulated shifts as the scale of the effect.
    shift = max(v[1] for v in emu.values())
    for precision in ("bf16",):
        gem = emu[precision][0]
        eng_b, _ = build("pairwise", S, precision=precision, coef=coef)
        ob = eng_b.step_body(batch, ds, teacher_labels=eng_b.labels_from_rows(ref["per_teacher"], A))
        torch.cuda.synchronize()
        ghip = eng_b.student.ps.export_grads()
        c_emu, c_hip, c_he = cosine(gem, g32), cosine(ghip, g32), cosine(ghip, gem)
        _D4_EMU[precision] = (c_emu, shift)
        print("D4/768 %s step: gradient cos emulation vs fp32 %.4f, HIP vs fp32 %.4f, HIP vs emulation %.4f; loss shift of the emulation %.2e, of HIP cls %.2e reg %.2e" % (
            precision, c_emu, c_hip, c_he, emu[precision][1], abs(ob["cls"].item() - loss32[1]) / abs(loss32[1]), abs(ob["reg"].item() - loss32[0]) / abs(loss32[0])))
        assert (1 - c_hip) <= 2.0 * (1 - c_emu) + 2e-2 and (1 - c_he) <= 2.0 * (1 - c_emu) + 2e-2, (precision, c_emu, c_hip, c_he)
        # (with cos_emu ~ 0.4 the angle bound above is loose; what separates a working mode from an indexing bug - uncorrelated gradients,
        # cos ~ 0 - is that the HIP gradient is as aligned with fp32 and with the emulation as the emulation is with fp32: measured ratios
        # 1.02 - 1.17 and 1.4 - 1.5)
        assert c_hip >= 0.75 * c_emu and c_he >= 0.75 * c_emu, (precision, c_emu, c_hip, c_he)
        assert abs(ob["reg"].item() - loss32[0]) <= (2.0 * shift + 2e-2) * abs(loss32[0]) and abs(ob["cls"].item() - loss32[1]) <= (2.0 * shift + 2e-2) * abs(loss32[1])
        np.testing.assert_allclose(ob["kd"].cpu().numpy(), loss32[2], rtol=0.1, atol=1e-3)
        del eng_b
        torch.cuda.empty_cache()


def test_graph_variants_plain_and_list_augmented():
    """traditional_nms_kdlist_augmented alternates, iteration by iteration, between the plain KD-list step and the one with the extra
    RGB-teacher pass: one set of hipGraphs per variant, picked by the batch's keys; each replays its own eager step."""
    S, B = 128, 2
    eng, spec = build("list", S)
    ref, _ = build("list", S)
    plain = {k: v.to(DEV) for k, v in synth_inputs(B, S, seed=5).items()}
    aug = dict(plain, aug_rgb=synth_inputs(B, S, seed=6)["rgb"].to(DEV))
    ds = eng.make_drop_scale(B, torch.Generator(device=DEV).manual_seed(1))
    def sync_state():      # both engines start every step from the same parameters / running statistics / optimizer state, so a step is
        # compared like a first step (two free-running engines drift apart by up to 2 lr per weight and step - Adam's first steps are
        # sign-like - and train-mode BatchNorm over 2 x 128^2 amplifies that to a few % of the classification loss)
        for dst, src in ((eng.student.ps, ref.student.ps),):
            for name in ("flat", "rmean", "rvar", "nbt"):
                getattr(dst, name).copy_(getattr(src, name))
        for name in ("exp_avg", "exp_avg_sq", "adam_main", "adam_head", "head_active"):
            getattr(eng, name).copy_(getattr(ref, name))
        eng.student.refresh()

    for batch in (plain, aug, plain, aug):
        sync_state()
        o = eng.replay(batch, ds)
        r = ref.step(batch, ds)
        torch.cuda.synchronize()
        assert o["nbox"].tolist() == r["nbox"].tolist()
        np.testing.assert_allclose(o["kd"].cpu().numpy(), r["kd"].cpu().numpy(), rtol=2e-3, atol=1e-5)
        np.testing.assert_allclose(o["cls"].cpu().numpy(), r["cls"].cpu().numpy(), rtol=2e-3)
        assert (eng.student.ps.flat - ref.student.ps.flat).abs().max().item() <= 2.5e-4      # one Adam step of <= lr each, from equal states
    assert set(eng._graphs) == {"plain", "aug"}


def test_graph_replay_matches_eager():
    """The captured hipGraphs must reproduce the eager step: same pseudo-labels and losses, gradients equal up to the
    run-to-run noise of fp32 atomics (measured eager-vs-eager: ~2e-4 of the largest gradient at this tiny size), and
    every replay must start from cleared accumulators (labels of a frozen teacher cannot change between replays)."""
    S, B = 128, 2
    batch = {k: v.to(DEV) for k, v in synth_inputs(B, S, seed=5).items()}
    eng_a, spec = build("pairwise", S)
    eng_b, _ = build("pairwise", S)
    g = torch.Generator(device=DEV).manual_seed(1)
    ds = eng_a.make_drop_scale(B, g)
    eng_b.capture(batch)
    oa = eng_a.step_body(batch, ds)
    eng_b.set_drop_scale(ds)
    eng_b.g_main.replay()
    torch.cuda.synchronize()
    ob = eng_b.out
    assert oa["nbox"].tolist() == ob["nbox"].tolist()
    for k in ("reg", "cls", "kd"):
        np.testing.assert_allclose(oa[k].cpu().numpy(), ob[k].cpu().numpy(), rtol=1e-4, atol=1e-6)
    ga, gb = eng_a.student.ps.grad, eng_b.student.ps.grad
    assert torch.isfinite(gb).all()
    assert (ga - gb).abs().max().item() <= 2e-3 * ga.abs().max().item()
    eng_a.optimizer_body(); eng_b.g_opt.replay()
    torch.cuda.synchronize()
    assert eng_a.adam_main[0].item() == eng_b.adam_main[0].item() == 1.0
    assert torch.equal(eng_a.student.ps.nbt, eng_b.student.ps.nbt)
    # replays 2 and 3: frozen teachers + same inputs -> identical labels every time
    for _ in range(2):
        eng_b.replay(batch, ds)
    torch.cuda.synchronize()
    assert eng_b.out["nbox"].tolist() == oa["nbox"].tolist()
    assert torch.isfinite(eng_b.student.ps.flat).all() and eng_b.adam_main[0].item() == 3.0


def test_split_backward_matches_unsplit():
    """The data-parallel step issues the backward in two segments (heads + BiFPN + backbone blocks >= k, then the early
    blocks + stem) so that the all-reduce of the first segment's gradients overlaps the second.  Same kernels, same order:
    gradients equal the unsplit backward's up to fp32-atomics noise, eagerly and through the three captured graphs."""
    S, B = 128, 2
    batch = {k: v.to(DEV) for k, v in synth_inputs(B, S, seed=5).items()}
    eng_a, spec = build("pairwise", S)
    eng_b, _ = build("pairwise", S)
    eng_c, _ = build("pairwise", S)
    k = eng_b._default_split()
    assert 0 < k < len(spec.blocks)
    eng_b.ar_split = eng_c.ar_split = k
    (p0,), tail = eng_b.grad_buckets()
    n = eng_b.student.ps.n_params
    assert sorted([p0] + tail) == [(0, p0[0]), p0, (p0[1], n)]            # the three ranges tile the buffer
    assert (p0[1] - p0[0]) > 0.9 * n
    g = torch.Generator(device=DEV).manual_seed(1)
    ds = eng_a.make_drop_scale(B, g)
    eng_a.step_body(batch, ds)
    eng_b.step_body(batch, ds)
    torch.cuda.synchronize()
    ga, gb = eng_a.student.ps.grad, eng_b.student.ps.grad
    tol = 2e-3 * ga.abs().max().item()
    # after segment 1 the overlapped bucket is final, the early blocks' weights have no gradient yet
    assert (ga[p0[0]:p0[1]] - gb[p0[0]:p0[1]]).abs().max().item() <= tol
    assert gb[:p0[0]].abs().max().item() == 0.0
    eng_b.backward_tail()
    torch.cuda.synchronize()
    assert (ga - gb).abs().max().item() <= tol
    eng_c.capture(batch)
    assert eng_c.g_tail is not None
    eng_c.replay(batch, ds)
    torch.cuda.synchronize()
    eng_a.optimizer_body()
    torch.cuda.synchronize()
    assert (ga - eng_c.student.ps.grad).abs().max().item() <= tol
    fa, fc = eng_a.student.ps.flat, eng_c.student.ps.flat
    assert torch.isfinite(fc).all()
    assert (fa - fc).abs().max().item() <= 2.5e-4            # one Adam step moves a weight by at most lr = 1e-4


@pytest.mark.parametrize("precision", ["bf16"])
def test_bf16_step_runs_and_replays(precision):
    """cfg `precision = bf16`: the whole distillation step (three teachers, student, losses, backward, Adam) with the 1x1
    convs on the bf16 MFMA.  At this test size (2 x 128^2, train-mode BatchNorm over a handful of samples) rounding noise
    is amplified far beyond what a real batch sees (tests/test_gpu_net.py), so only coarse agreement with the fp32 step
    is asserted; the exact statements are that the step is finite, trains (Adam moves every touched weight by ~lr) and
    that the captured graphs reproduce the eager bf16 step."""
    S, B = 128, 2
    batch = {k: v.to(DEV) for k, v in synth_inputs(B, S, seed=5).items()}
    eng_a, spec = build("pairwise", S)
    eng_b, _ = build("pairwise", S, precision=precision)
    eng_c, _ = build("pairwise", S, precision=precision)
    assert eng_b.student._sfx == "_bf16" and all(t._sfx == "_bf16" for t in eng_b.teachers.values())
    g = torch.Generator(device=DEV).manual_seed(1)
    ds = eng_a.make_drop_scale(B, g)
    p0 = eng_b.student.ps.flat.clone()
    oa = eng_a.step(batch, ds)
    ob = eng_b.step(batch, ds)
    torch.cuda.synchronize()
    kd_a, kd_b = oa["kd"].cpu().numpy(), ob["kd"].cpu().numpy()
    assert np.isfinite(kd_b).all()
    assert not np.array_equal(kd_a, kd_b)                       # the bf16 kernels really ran
    np.testing.assert_allclose(kd_b.sum(), kd_a.sum(), rtol=0.5)
    db = eng_b.student.ps.flat - p0
    assert torch.isfinite(db).all() and 0.5e-4 < db.abs().max().item() < 2.5e-4
    eng_c.capture(batch)
    eng_c.replay(batch, ds)
    torch.cuda.synchronize()
    oc = eng_c.out
    assert oc["nbox"].tolist() == ob["nbox"].tolist()
    np.testing.assert_allclose(oc["kd"].cpu().numpy(), kd_b, rtol=1e-4, atol=1e-7)
    assert (eng_c.student.ps.flat - eng_b.student.ps.flat).abs().max().item() <= 2.5e-4


@pytest.mark.parametrize("teach", ["rgb+depth+thermal", "rgb"])      # BASELINE configs[2] (the headline) and configs[1] (one RGB teacher)
def test_full_size_step_graph_vs_oracle(teach):
    """The HEADLINE workload end to end (BASELINE configs[2]: three frozen D2 teachers + the audio student, 512 x 512, per-GPU batch 8), the
    way bench.py runs it - DistillEngine.capture() + two replay()s - against oracle/step_ref on the same inputs, weights and drop-connect
    masks: teachers on staggered side streams, decode + NMS + cross-teacher merge at ~60 candidates per teacher and image, focal loss over
    A = 49 104 anchors, the MTA terms, the student's backward incl. the grouped weight gradients, Adam, and the same again from the updated
    weights.  Reference: src/optimization/train_methods.py:436-517, src/optimization/traditional.py:171-190."""
    import bench as BN
    from oracle import step_ref as ST
    from helpers import grad_state
    S, B, coef = 512, 8, 2
    from mm_distillnet_amd.hostinfo import free_memory_gb
    free_gb = free_memory_gb()          # (machine-available and cgroup limit)
    if free_gb < 64:
        pytest.skip(f"full-size step test needs >= 64 GB of free host memory for the oracle's autograd tape (have {free_gb:.0f} GB): NOT RUN at B = 8")
    print(f"full-size whole-step parity test: B = {B}, {S} x {S} ({free_gb:.0f} GB of host memory free)")
    hb = synth_inputs(B, S, seed=41)
    teachers = teacher_states(coef, {k: MODS[k] for k in teach.split("+")})
    NT = len(teachers)
    for k, (spec_t, st_t) in teachers.items():      # ~60 over-threshold candidates per image and teacher (bench.py's recipe)
        BN.tune_teacher_bias(spec_t, st_t, hb[k], DEV, target_per_image=60)
    spec_s, st_s = make_state(coef, 8, 24, "audio")
    tstates = {k: v[1] for k, v in teachers.items()}

    def engine():
        e = DistillEngine(spec_s, {k: v[0] for k, v in teachers.items()}, DEV, StepConfig(image_size=S))
        e.load({k: v.clone() for k, v in st_s.items()}, tstates)
        return e

    skip = [b for b in spec_s.blocks if b.skip]
    gen = torch.Generator().manual_seed(7)
    masks = [{b.idx: torch.floor((1.0 - b.drop_rate) + torch.rand(B, generator=gen)) for b in skip} for _ in range(2)]
    ds = [torch.stack([m[b.idx] / (1.0 - b.drop_rate) for b in skip]).to(DEV) for m in masks]
    batch = {k: v.to(DEV) for k, v in hb.items()}
    # ---- oracle: one step (forward, losses, backward, Adam) from its current state; called once per GPU step below
    so = grad_state(st_s)
    params = {k: v for k, v in so.items() if v.requires_grad}
    opt_state = {}

    def oracle_step(step):
        for v in params.values():
            v.grad = None
        ref = ST.distill_forward(so, tstates, hb, S, coef, masks[step])
        ST.total_loss(ref).backward()
        grads = {k: v.grad.detach().clone() for k, v in params.items() if v.grad is not None}
        with torch.no_grad():
            ST.adam_step(params, grads, opt_state)
        return {"reg": ref["reg"].item(), "cls": ref["cls"].item(), "kd": torch.stack(ref["kd"]).detach().numpy(), "grads": grads,
                "per_teacher": ref["per_teacher"], "labels": ref["labels"], "weights": {k: v.detach().clone() for k, v in params.items()}}

    refs = [oracle_step(0)]
    nlab = [int(np.size(l) // 6) for t in refs[0]["per_teacher"] for l in t]
    print("oracle pseudo-labels per (teacher, image):", nlab, "merged per image:", [int(np.size(l) // 5) for l in refs[0]["labels"]])
    assert sum(nlab) >= NT * B * 10

    def compare(grads, ref_grads):
        dot = n1 = n2 = 0.0
        for k, a in ref_grads.items():
            a, b_ = a.double(), grads[k].double()
            dot += float((a * b_).sum()); n1 += float((a * a).sum()); n2 += float((b_ * b_).sum())
        return dot / (n1 ** 0.5 * n2 ** 0.5), (n2 / n1) ** 0.5

    # ---- (1) eager step with the oracle's pseudo-labels: everything downstream of the labels at kernel tolerances
    eng = engine()
    A = eng.student.anchors(S).shape[0]
    assert A == 49104
    out = eng.step_body(batch, ds[0], teacher_labels=eng.labels_from_rows(refs[0]["per_teacher"], A))
    torch.cuda.synchronize()
    eng.check_overflow()
    np.testing.assert_allclose(out["reg"].item(), refs[0]["reg"], rtol=2e-4)
    np.testing.assert_allclose(out["cls"].item(), refs[0]["cls"], rtol=2e-4)
    np.testing.assert_allclose(out["kd"].cpu().numpy(), refs[0]["kd"].reshape(out["kd"].shape), rtol=1e-4, atol=1e-5)
    cos, ratio = compare(eng.student.ps.export_grads(), refs[0]["grads"])
    print("full-size step, oracle labels: gradient cos %.6f norm ratio %.5f" % (cos, ratio))
    assert cos >= 0.9999 and abs(ratio - 1.0) < 2e-3, (cos, ratio)
    del eng
    torch.cuda.empty_cache()
    # ---- (2) the bench path: capture + replay, the GPU teachers' own labels
    eng = engine()
    eng.capture(batch)
    lr = eng.lr
    for step in range(2):
        if step == 1:
            # the second step is compared from IDENTICAL states: the oracle continues from the HIP engine's parameters and running
            # statistics after its first step (Adam's first step is sign-like, so the two trajectories differ by up to 2 lr on the few
            # weights whose gradient is ~eps, and these random-weight nets amplify that to ~1 % of the next gradient)
            w_gpu = eng.student.ps.export_state()
            with torch.no_grad():
                for k, v in so.items():
                    if v.dtype.is_floating_point:
                        v.copy_(w_gpu[k])
            refs.append(oracle_step(1))
        out = eng.replay(batch, ds[step])
        torch.cuda.synchronize()
        eng.check_overflow()
        r = refs[step]
        tot = hit = 0
        for ti in range(NT):
            for i in range(B):
                rr = np.asarray(r["per_teacher"][ti][i], dtype=np.float32).reshape(-1, 6)
                n = int(out["cnt_t"][ti][i].item())
                got = out["rows_t"][ti][i, :n].cpu().numpy()
                assert abs(n - rr.shape[0]) <= max(2, 0.05 * rr.shape[0]), (step, ti, i, n, rr.shape[0])
                for row in rr:
                    tot += 1
                    hit += int(n > 0 and (np.abs(got[:, :4] - row[:4]).max(1) <= 1.0).any())
        assert hit >= 0.95 * tot, (step, hit, tot)
        nb = out["nbox"].cpu().tolist()
        same = all(np.array_equal(out["boxes"][i, :nb[i]].cpu().numpy(), np.asarray(r["labels"][i], dtype=np.float32).reshape(-1, 5)) for i in range(B))
        tol = 2e-4 if same else 2e-2
        print("full-size step %d (graph replay): %d / %d teacher rows within 1 px, merged labels %s -> loss tolerance %g; reg %.6f (%.6f) cls %.6f (%.6f)"
              % (step, hit, tot, "identical" if same else "differ by integer truncation", tol, out["reg"].item(), r["reg"], out["cls"].item(), r["cls"]))
        np.testing.assert_allclose(out["reg"].item(), r["reg"], rtol=tol)
        np.testing.assert_allclose(out["cls"].item(), r["cls"], rtol=tol)
        np.testing.assert_allclose(out["kd"].cpu().numpy(), r["kd"].reshape(out["kd"].shape), rtol=1e-4 if step == 0 else 2e-3, atol=1e-5)
        cos, ratio = compare(eng.student.ps.export_grads(), r["grads"])
        print("full-size step %d (graph replay): gradient cos %.6f norm ratio %.5f" % (step, cos, ratio))
        # Step 0 (identical initial state on both sides) holds the tight bound.  Step 1 starts from the state the FIRST Adam step left - sign-like
        # for |g| ~ eps, so it differs from run to run at the 1e-3 level (the oracle follows the engine's exported state, but the point the
        # gradient is evaluated at moves: reg / cls of step 1 vary in the 4th digit between runs) - and its gradient cosine was measured between
        # 0.99959 and 1.000000 over 20 runs of the round-6 build (median 0.99999; norm ratio within 1.1e-3): bound 0.999 / 5e-3.
        tight = same and step == 0
        assert cos >= (0.9999 if tight else 0.999) and abs(ratio - 1.0) < (2e-3 if tight else (5e-3 if same else 2e-2)), (step, cos, ratio)
        # Adam-updated weights: every step moves a weight by <= lr (sign-like while |g| ~ eps); with equal gradients the two
        # trajectories stay within a small fraction of that
        w = eng.student.ps.export_state()
        worst, close_frac, n_el = 0.0, 0.0, 0
        for k, a in r["weights"].items():
            dlt = (w[k].double() - a.double()).abs()
            worst = max(worst, float(dlt.max()))
            close_frac += float((dlt <= 0.05 * lr * (step + 1)).sum()); n_el += dlt.numel()
        print("full-size step %d: Adam-updated weights max |diff| %.2e (lr %.0e), %.4f of the elements within 5 %% of lr" % (step, worst, lr, close_frac / n_el))
        # (worst case: an element whose gradient is ~eps gets +lr on one side and -lr on the other)
        assert worst <= 2.05 * lr * (step + 1) and close_frac / n_el >= (0.99 if same else 0.95)
    assert eng.adam_main[0].item() == 2.0


# ---- D4 step against the reference's own ModelWithNMSLoss built from D4 nets (tools/oracle/make_golden.py golden_step_d4)
D4_STEP_MODS = {"rgb": (3, 51), "depth": (3, 52), "thermal": (1, 53)}


def build_d4_golden(S=256):
    teachers = {k: make_state(4, cin, seed, k, cls_bias=-2.0) for k, (cin, seed) in D4_STEP_MODS.items()}
    spec_s, st_s = make_state(4, 8, 54, "audio")
    eng = DistillEngine(spec_s, {k: v[0] for k, v in teachers.items()}, DEV, StepConfig(image_size=S))
    eng.load(st_s, {k: v[1] for k, v in teachers.items()})
    return eng, spec_s


@pytest.mark.parametrize("own_labels", [False, True])
def test_step_d4_golden(golden_dir, own_labels):
    """BASELINE configs[4]'s architecture through the whole step, against a fixture made by the REFERENCE's D4 classes (not by the
    oracle): 3 frozen D4 teachers + the D4 student at 256^2, B = 2.  own_labels False: the teachers' pseudo-label rows come from the
    reference run - losses 2e-4, gradients 2e-3, Adam 1e-5 as for D2; True: the GPU teachers' own decode + NMS - rows within 1 px,
    losses 2e-2."""
    gold = np.load(os.path.join(golden_dir, "step_d4_256_pairwise.npz"))
    S, B = 256, 2
    eng, spec = build_d4_golden(S)
    batch = {k: v.to(DEV) for k, v in synth_inputs(B, S, seed=71).items()}
    ds = drop_scale_from(gold, spec)
    if own_labels:
        out = eng.step_body(batch, ds)
    else:
        A = eng.student.anchors(S).shape[0]
        labels = eng.labels_from_rows([[gold[f"teacher{ti}_img{i}"] for i in range(B)] for ti in range(3)], A)
        out = eng.step_body(batch, ds, teacher_labels=labels)
    torch.cuda.synchronize()
    eng.check_overflow()
    if own_labels:
        tot = hit = 0
        for ti in range(3):
            for i in range(B):
                ref = gold[f"teacher{ti}_img{i}"]
                n = int(out["cnt_t"][ti][i].item())
                got = out["rows_t"][ti][i, :n].cpu().numpy()
                assert abs(n - ref.shape[0]) <= max(2, 0.05 * ref.shape[0]), (ti, i, n, ref.shape[0])
                for r in ref:
                    tot += 1
                    hit += bool(n and (np.abs(got[:, :4] - r[:4]).max(1) <= 1.0).any())
        assert hit >= 0.95 * tot, (hit, tot)
    lt = 2e-2 if own_labels else 2e-4
    np.testing.assert_allclose(out["reg"].cpu().numpy(), gold["reg"], rtol=lt)
    np.testing.assert_allclose(out["cls"].cpu().numpy(), gold["cls"], rtol=lt)
    np.testing.assert_allclose(out["kd"].cpu().numpy(), gold["kd"].reshape(out["kd"].shape), rtol=1e-4, atol=1e-5)
    loss = 1.0 * (out["reg"].item() + out["cls"].item()) + 0.005 * out["kd"].sum().item()
    assert abs(loss - float(gold["loss"])) < lt * abs(float(gold["loss"]))
    if own_labels:
        # (the stem's gradient - the far end of a 32-block, 7-cell deep chain with BatchNorm over 8 samples on the 2 x 2 level - measured
        # 4e-3 of its largest element from the reference run with the GPU teachers' own labels, 2e-3 with the reference's)
        grad_checks(gold, eng.student.ps.export_grads(), 2e-2, 3e-2, 1e-2)
    else:
        # (run to run the D4 student's watched gradients sit 2e-3 .. 6e-3 of their largest element from the reference run - the trainable
        # net's atomics order, amplified by 32 blocks + 7 cells with BatchNorm over 8 samples on the 2 x 2 level: tests/test_gpu_net.py
        # test_net_d4_train_fwd_bwd_golden; the gradient norms per module hold 1e-2, the losses 2e-4)
        grad_checks(gold, eng.student.ps.export_grads(), 1e-2, 2e-2, 1e-2, fusion_atol=1e-2)
    eng.optimizer_body()
    torch.cuda.synchronize()
    params = eng.student.ps.export_state()
    if not own_labels:
        for k in gold.files:
            if k.startswith("adam.") and k.endswith(".head"):
                name = k[5:-5]
                check_summary(gold, "adam." + name, params[name], 1e-5, 1e-4)
