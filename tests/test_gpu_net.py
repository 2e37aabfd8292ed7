"""GPU parity of the whole detector on the HIP engine against the oracle and the reference golden vectors.
Tolerance (fp32, stated in SURVEY.md §8c-5): outputs rtol 1e-3 / atol 1e-4 of the tensor scale; gradients
rtol 1e-2 on norms, 2e-2 on sampled heads (fp32 atomics reorder sums)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from mm_distillnet_amd.arch import make_spec
from mm_distillnet_amd.engine import Net
from mm_distillnet_amd.synth import synth_inputs
from oracle import effdet_ref as O
from helpers import make_state, check_summary, grad_state

DEV = "cuda"


def feat_nchw(f):
    return f.z.view(f.B, f.H, f.W, f.C).permute(0, 3, 1, 2)


def relerr(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return (a - b).abs().max().item() / max(b.abs().max().item(), 1e-30)


@pytest.mark.parametrize("mod,cin,seed", [("rgb", 3, 11), ("thermal", 1, 12), ("audio", 8, 13)])
def test_net_eval_golden(golden_dir, mod, cin, seed):
    gold = np.load(os.path.join(golden_dir, f"net_d2_eval_{mod}.npz"))
    spec, st = make_state(2, cin, seed, mod)
    net = Net(spec, DEV, trainable=False)
    net.load_state(st)
    x = synth_inputs(2, 128, seed=24)[mod]
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=False)
    torch.cuda.synchronize()
    with torch.no_grad():
        (c, r, a), f = O.forward(st, x, 2, False)
    assert relerr(cls, c) < 1e-3 and relerr(reg, r) < 1e-3
    for u, v in zip(feats, f):
        assert relerr(feat_nchw(u), v) < 1e-3
    assert torch.equal(net.anchors(128).cpu(), a[0])
    check_summary(gold, "cls", cls, 1e-3, 1e-4); check_summary(gold, "reg", reg, 1e-3, 1e-4)
    for i, u in enumerate(feats):
        check_summary(gold, f"feat{i}", feat_nchw(u).contiguous(), 1e-3, 1e-4)
    # state round trip through the native layouts
    ex = net.ps.export_state()
    for k, v in st.items():
        assert torch.equal(ex[k].cpu().to(v.dtype), v), k


def _train_case(golden_dir):
    gold = np.load(os.path.join(golden_dir, "net_d2_train_audio.npz"))
    spec, st = make_state(2, 8, 13, "audio")
    x = synth_inputs(2, 128, seed=25)["audio"]
    masks = {int(b): torch.from_numpy(m) for b, m in zip(gold["drop_blocks"], gold["drop_masks"])}
    return gold, spec, st, x, masks


def test_net_train_fwd_bwd(golden_dir):
    gold, spec, st, x, masks = _train_case(golden_dir)
    # oracle
    so = grad_state(st)
    (c, r, a), f = O.forward(so, x, 2, True, masks)
    loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
    loss.backward()
    # engine
    net = Net(spec, DEV, trainable=True)
    net.load_state(st)
    skip = [b for b in spec.blocks if b.skip]
    ds = torch.stack([masks[b.idx] / (1.0 - b.drop_rate) for b in skip]).to(DEV)
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=True, drop_scale=ds)
    assert relerr(cls, c) < 1e-3 and relerr(reg, r) < 1e-3
    for u, v in zip(feats, f):
        assert relerr(feat_nchw(u), v) < 1e-3
    check_summary(gold, "cls", cls, 1e-3, 1e-4); check_summary(gold, "reg", reg, 1e-3, 1e-4)
    # same scalar loss: dL/dcls = 0.01 (prob) -> logit grad, dL/dreg = 2 r / numel, dL/dfeat = 2 f / numel
    dcls = (0.01 * cls * (1 - cls)).contiguous()
    dreg = (2.0 * reg / reg.numel()).contiguous()
    dfe = [(2.0 * u.z / u.z.numel()).contiguous() for u in feats]
    net.ps.grad.zero_()
    net.backward(dcls, dreg, dfe)
    torch.cuda.synchronize()
    grads = net.ps.export_grads()
    # running statistics
    ex = net.ps.export_state()
    for k in gold.files:
        if k.startswith("stat.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(gold, "stat." + name, ex[name], 1e-4, 1e-5)
    assert int(ex["backbone_net.model._bn0.num_batches_tracked"]) == int(gold["nbt"])
    # gradients: every parameter against the oracle's autograd (relative to the module's largest gradient)
    worst = []
    for k, v in so.items():
        if not v.requires_grad:
            continue
        ref = v.grad
        e = (grads[k].double() - ref.double()).abs().max().item()
        scale = max(ref.abs().max().item(), 1e-12)
        worst.append((e / scale, k, scale))
    worst.sort(reverse=True)
    gscale = max(s for _, _, s in worst)
    bad = [(e, k, s) for e, k, s in worst if e > 2e-2 and s > 1e-6 * gscale]
    assert not bad, bad[:10]
    for k in gold.files:
        if k.startswith("gradnorm."):
            top = k[len("gradnorm."):]
            tot = sum(float(grads[n].double().pow(2).sum()) for n in grads
                      if (".".join(n.split(".")[:2]) if n.startswith("bifpn") else n.split(".")[0]) == top)
            assert abs(tot ** 0.5 - float(gold[k])) <= 1e-2 * float(gold[k]), (top, tot ** 0.5, float(gold[k]))
        if k.startswith("grad.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(gold, "grad." + name, grads[name], 2e-2, 1e-3)


@pytest.mark.parametrize("B", [1, 2, 8])
def test_net_eval_512_vs_oracle(B):
    """BASELINE config-1 shape (one RGB teacher forward, 512x512) at B = 1, 2 and the benchmark's 8 against the oracle."""
    spec, st = make_state(2, 3, 11, "rgb")
    net = Net(spec, DEV, trainable=False)
    net.load_state(st)
    x = synth_inputs(B, 512, seed=3)["rgb"]
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=False)
    with torch.no_grad():
        (c, r, a), f = O.forward(st, x, 2, False)
    assert cls.shape == (B, 49104, 20) and reg.shape == (B, 49104, 4)
    assert relerr(cls, c) < 1e-3 and relerr(reg, r) < 1e-3
    for u, v in zip(feats, f):
        assert relerr(feat_nchw(u), v) < 1e-3


def _full_size_batch():
    """B = 8, the benchmark's per-GPU batch.  The oracle's autograd tape needs ~3 GB of host memory per image at 512^2: a host
    without it SKIPS (loudly) - the test never shrinks to a smaller batch on its own."""
    from mm_distillnet_amd.hostinfo import free_memory_gb
    free_gb = free_memory_gb()          # (machine-available and cgroup limit)
    if free_gb < 48:
        pytest.skip(f"full-size parity test needs >= 48 GB of free host memory for the oracle (have {free_gb:.0f} GB): NOT RUN at B = 8")
    print(f"full-size parity test: B = 8 ({free_gb:.0f} GB of host memory free)")
    return 8


def test_net_train_512_full_size_vs_oracle():
    """The student's TRAIN forward + hand-scheduled backward at the headline size (BASELINE configs[2]: 512 x 512, per-GPU batch 8)
    against the oracle's autograd.  At this size the dispatch takes the kernel variants the small goldens never reach: 128x32 /
    128x64 GEMM tiles with thousands of blocks and the XCD swizzle, BatchNorm sums spread over 64 workspace slots, the lean
    register variants, the slotted depthwise statistics, the split weight-gradient reductions.
    Bounds (fp32): outputs / features 1e-3 of the tensor's largest value; gradient direction cos > 0.9999 and norm within 2e-3
    over ALL parameters; per-tensor max error <= 2e-2 of the tensor's largest gradient for at least 95 % of the tensors and
    <= 6e-2 for every tensor - except the 2-/3-element BiFPN fusion weights (p*_w1 / p*_w2), whose gradient is one dot product
    over a whole level and moves by up to ~15 % when a max-pool tie on the 4x4 / 8x8 levels resolves the other way (measured
    0.149 on bifpn.0.p6_w1 with every other tensor <= 0.024): <= 0.25 for those."""
    B, S = _full_size_batch(), 512
    print("full-size train check at B =", B)
    spec, st = make_state(2, 8, 13, "audio")
    x = synth_inputs(B, S, seed=26)["audio"]
    g = torch.Generator().manual_seed(5)
    skip = [b for b in spec.blocks if b.skip]
    masks = {b.idx: torch.floor((1.0 - b.drop_rate) + torch.rand(B, generator=g)) for b in skip}
    so = grad_state(st)
    (c, r, a), f = O.forward(so, x, 2, True, masks)
    loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
    loss.backward()
    c, r, f = c.detach(), r.detach(), [u.detach() for u in f]
    net = Net(spec, DEV, trainable=True)
    net.load_state(st)
    ds = torch.stack([masks[b.idx] / (1.0 - b.drop_rate) for b in skip]).to(DEV)
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=True, drop_scale=ds)
    assert cls.shape == (B, 49104, 20)
    assert relerr(cls, c) < 1e-3 and relerr(reg, r) < 1e-3, (relerr(cls, c), relerr(reg, r))
    for u, v in zip(feats, f):
        assert relerr(feat_nchw(u), v) < 1e-3
    dcls = (0.01 * cls * (1 - cls)).contiguous()
    dreg = (2.0 * reg / reg.numel()).contiguous()
    dfe = [(2.0 * u.z / u.z.numel()).contiguous() for u in feats]
    net.ps.grad.zero_()
    net.backward(dcls, dreg, dfe)
    torch.cuda.synchronize()
    grads = net.ps.export_grads()
    ex = net.ps.export_state()
    for k in ("backbone_net.model._bn0.running_mean", "backbone_net.model._blocks.10._bn1.running_var",
              "bifpn.3.conv4_up.bn.running_var", "regressor.bn_list.4.0.running_mean"):
        assert relerr(ex[k], so[k]) < 1e-4, k          # the oracle updated its copy of the running statistics in place
    dot = n1 = n2 = 0.0
    errs = []
    gmax = max(v.grad.abs().max().item() for v in so.values() if v.requires_grad)
    for k, v in so.items():
        if not v.requires_grad:
            continue
        ref, got = v.grad.double(), grads[k].double()
        dot += float((ref * got).sum()); n1 += float((ref * ref).sum()); n2 += float((got * got).sum())
        s_ = ref.abs().max().item()
        if s_ > 1e-5 * gmax:
            errs.append(((got - ref).abs().max().item() / s_, k))
    cos, ratio = dot / (n1 ** 0.5 * n2 ** 0.5), (n2 / n1) ** 0.5
    errs.sort()
    print("512^2 train gradient check: cos %.7f norm ratio %.5f per-tensor max rel err p50 %.2e p95 %.2e max %.2e (%s)" % (
        cos, ratio, errs[len(errs) // 2][0], errs[int(0.95 * len(errs))][0], errs[-1][0], errs[-1][1]))
    assert cos > 0.9999 and abs(ratio - 1.0) < 2e-3, (cos, ratio)
    is_fuse = lambda k: k.split(".")[-1] in ("p6_w1", "p5_w1", "p4_w1", "p3_w1", "p4_w2", "p5_w2", "p6_w2", "p7_w2")
    rest = [e for e in errs if not is_fuse(e[1])]
    assert errs[int(0.95 * len(errs))][0] < 2e-2 and rest[-1][0] < 6e-2, errs[-8:]
    # the 2- / 3-element fast-attention weights: d theta_k = sum_i wdot_i (delta_ik S - r_i) / S^2 is a difference of large dot products,
    # so one weight's gradient can be tiny next to its own rounding noise (per-tensor relative error 0.15-0.26 run to run, fp32 atomics
    # in the dot products); measured against the largest fusion-weight gradient of the net the error is stable
    fk = [k for k, v in so.items() if v.requires_grad and is_fuse(k)]
    fmax = max(so[k].grad.abs().max().item() for k in fk)
    ferr = max((grads[k].double() - so[k].grad.double()).abs().max().item() for k in fk) / fmax
    print("fusion weights: max abs err / largest fusion-weight gradient %.3e" % ferr)
    assert ferr < 5e-3, ferr          # measured 1.0-1.1e-3 in every run


def test_d4_eval_vs_oracle():
    """BASELINE config-5 architecture (EfficientDet-D4: b4 backbone, 7 BiFPN cells of width 224, 4-layer heads) in fp32
    on a 256x256 input against the oracle (the reference's load_model hard-codes D2; the classes support D4)."""
    spec, st = make_state(4, 3, 31, "rgb")
    net = Net(spec, DEV, trainable=False)
    net.load_state(st)
    x = synth_inputs(2, 256, seed=8)["rgb"]
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=False)
    with torch.no_grad():
        (c, r, a), f = O.forward(st, x, 4, False)
    assert cls.shape == c.shape and reg.shape == r.shape
    assert relerr(cls, c) < 2e-3 and relerr(reg, r) < 2e-3
    for u, v in zip(feats, f):
        assert relerr(feat_nchw(u), v) < 2e-3


def test_d4_768_full_batch_eval_vs_oracle():
    """BASELINE configs[4]'s frozen net at its full per-GPU size (D4, 8 x 768², fp32): against the oracle, and - the size-independent
    property of an eval-mode net - every image of the batch equal to its own batch-of-one forward.  (Not bit for bit: a batch of one takes
    other tile variants of the same kernels and the squeeze-excite pool's atomics add in another order; the features agree to ~1e-5 and
    this synthetic classifier head turns that into a few 1e-4 of probability, the same amplification the oracle comparison shows.)"""
    spec, st = make_state(4, 3, 31, "rgb")
    net = Net(spec, DEV, trainable=False)
    net.load_state(st)
    B, S = 8, 768
    x = synth_inputs(B, S, seed=12)["rgb"]
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=False)
    cls, reg, feats = cls.clone(), reg.clone(), [feat_nchw(u).clone() for u in feats]
    with torch.no_grad():
        (c, r, a), f = O.forward(st, x, 4, False)
    assert cls.shape == c.shape and reg.shape == r.shape
    errs = (relerr(cls, c), relerr(reg, r), max(relerr(u, v) for u, v in zip(feats, f)))
    print("D4 8 x 768² eval vs oracle: max error / tensor max - cls %.1e reg %.1e features %.1e" % errs)
    assert max(errs) < 2e-3, errs
    for i in (0, B - 1):
        net.begin_step()
        c1, r1, _ = net.forward(x[i:i + 1].to(DEV), train=False)
        e_c, e_r = relerr(c1[0], cls[i]), relerr(r1[0], reg[i])
        print("   image %d alone vs inside the batch: cls %.1e reg %.1e" % (i, e_c, e_r))
        assert e_c < 2e-3 and e_r < 1e-4, (i, e_c, e_r)


@pytest.mark.parametrize("coef,size", [(0, 256), (1, 384), (3, 256)])      # input sizes are multiples of 128, as upstream (nn.Upsample x2 between levels)
def test_other_compound_coefficients_eval_vs_oracle(coef, size):
    """EfficientDet-D0 / D1 / D3 frozen nets against the oracle: widths for which the fused frozen-net kernels have no instantiation
    (BiFPN width 64 / 88 / 160: two-kernel node path) next to blocks that do use
    them, and the row-streaming depthwise kernel at other channel counts."""
    spec, st = make_state(coef, 3, 40 + coef, "rgb")
    net = Net(spec, DEV, trainable=False)
    net.load_state(st)
    x = synth_inputs(2, size, seed=9)["rgb"]
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=False)
    with torch.no_grad():
        (c, r, a), f = O.forward(st, x, coef, False)
    assert cls.shape == c.shape and reg.shape == r.shape
    assert relerr(cls, c) < 2e-3 and relerr(reg, r) < 2e-3
    for u, v in zip(feats, f):
        assert relerr(feat_nchw(u), v) < 2e-3


def test_d4_train_fwd_bwd_vs_oracle():
    """BASELINE config-5 architecture (D4, 8-channel student) through the TRAIN forward and the hand-scheduled backward in
    fp32 against the oracle's autograd: 7 BiFPN cells of width 224, 4-layer heads, 48-channel stem, 32 MBConv blocks."""
    spec, st = make_state(4, 8, 32, "audio")
    x = synth_inputs(2, 256, seed=9)["audio"]
    masks = {b.idx: torch.ones(2) * (1.0 - b.drop_rate) for b in spec.blocks if b.skip}      # x/keep*mask with mask = keep -> x
    so = grad_state(st)
    (c, r, a), f = O.forward(so, x, 4, True, masks)
    loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
    loss.backward()
    net = Net(spec, DEV, trainable=True)
    net.load_state(st)
    skip = [b for b in spec.blocks if b.skip]
    ds = torch.ones(len(skip), 2, device=DEV)
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=True, drop_scale=ds)
    assert relerr(cls, c) < 2e-3 and relerr(reg, r) < 2e-3
    for u, v in zip(feats, f):
        assert relerr(feat_nchw(u), v) < 2e-3
    dcls = (0.01 * cls * (1 - cls)).contiguous()
    dreg = (2.0 * reg / reg.numel()).contiguous()
    dfe = [(2.0 * u.z / u.z.numel()).contiguous() for u in feats]
    net.ps.grad.zero_()
    net.backward(dcls, dreg, dfe)
    torch.cuda.synchronize()
    grads = net.ps.export_grads()
    # whole gradient: direction and norm; element-wise: all but a few tensors (max-pool ties can flip, see test_gpu_model.py)
    dot = n1 = n2 = 0.0
    errs = []
    gmax = max(v.grad.abs().max().item() for v in so.values() if v.requires_grad)
    for k, v in so.items():
        if not v.requires_grad:
            continue
        ref, got = v.grad.double(), grads[k].double()
        dot += float((ref * got).sum()); n1 += float((ref * ref).sum()); n2 += float((got * got).sum())
        s = ref.abs().max().item()
        if s > 1e-4 * gmax:
            errs.append((got - ref).abs().max().item() / s)
    assert dot / (n1 ** 0.5 * n2 ** 0.5) > 0.9995 and abs((n2 / n1) ** 0.5 - 1.0) < 5e-3
    errs.sort()
    print("d4 train gradient check: per-tensor max relative error p90 %.4f p95 %.4f max %.4f" % (
        errs[int(0.90 * len(errs))], errs[int(0.95 * len(errs))], errs[-1]))
    # fp32 atomics + max-pool ties make this statistic bimodal from run to run: p95 = 0.003 / 0.009 / 0.021 in three runs of the
    # SAME build (and 0.003 / 0.008 / 0.021 with MMD_NO_LAZY_BN=1, the separate BatchNorm apply pass) - a flipped tie reroutes a
    # gradient through the tiny 2x2 / 4x4 levels of this 256x256 test; the direction / norm checks above are the tight ones
    assert errs[int(0.95 * len(errs))] < 3e-2, errs[-10:]


def l2rel(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return ((a - b).norm() / b.norm().clamp_min(1e-30)).item()


def _bf16_oracle(fn):
    O.BF16_PW = True
    try:
        return fn()
    finally:
        O.BF16_PW = False


# bf16 operand rounding (2^-9 per operand) is amplified by these random-weight nets: the ORACLE's own bf16 emulation
# (oracle/effdet_ref.py BF16_PW: operands of the 1x1 convs rounded, fp32 accumulate) sits 3-6 % (max error / max value)
# from its fp32 result at D4 and D2.  Two implementations of the same rounding rule also decorrelate (a last-bit
# difference upstream flips roundings downstream), so the net-level bounds are statistical: RMS error against the fp32
# oracle, and "no farther from the emulation than the emulation is from fp32".  The tight checks are the op-level ones
# (tests/test_gpu_kernels.py::test_pwconv_bf16_*: equal to an fp32 GEMM of the rounded operands).
BF16_RMS = 5e-2


@pytest.mark.parametrize("precision", ["bf16"])
def test_d4_eval_bf16_vs_oracle(precision):
    """BASELINE config 5 (D4, bf16 mixed precision): the 1x1-conv GEMMs run on the bf16 MFMA (operands rounded, fp32
    accumulate); everything else stays fp32.  (The bf16 STORAGE mode "bf16_hbm" was deleted in round 6.)"""
    spec, st = make_state(4, 3, 31, "rgb")
    x = synth_inputs(2, 256, seed=8)["rgb"]
    net = Net(spec, DEV, trainable=False, precision=precision)
    net.load_state(st)
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=False)
    net32 = Net(spec, DEV, trainable=False)
    net32.load_state(st)
    net32.begin_step()
    cls32, reg32, _ = net32.forward(x.to(DEV), train=False)
    with torch.no_grad():
        (c, r, a), f = O.forward(st, x, 4, False)
        (cb, rb, _), fb = _bf16_oracle(lambda: O.forward(st, x, 4, False))
    print("D4 eval %s: HIP vs fp32 oracle %.4f (reg) / emulation vs fp32 %.4f / HIP vs emulation %.4f" % (precision, l2rel(reg, r), l2rel(rb, r), l2rel(reg, rb)))
    assert l2rel(reg, r) < BF16_RMS and l2rel(cls, c) < 3 * BF16_RMS, (l2rel(reg, r), l2rel(cls, c))
    for u, v, w in zip(feats, f, fb):
        assert l2rel(feat_nchw(u), v) < BF16_RMS
        assert l2rel(feat_nchw(u), w) < 2 * l2rel(w, v) + 1e-3
    assert l2rel(reg, rb) < 2 * l2rel(rb, r) + 1e-3
    assert relerr(reg, reg32) > 1e-5        # the bf16 kernels really ran


@pytest.mark.parametrize("precision", ["bf16"])
def test_d2_train_bf16_fwd_bwd_vs_oracle(precision):
    """Train forward + hand-scheduled backward of the 8-channel D2 student with bf16 GEMMs (forward, input- and
    weight-gradient).  Train-mode BatchNorm over the few samples of a test-sized batch amplifies rounding noise (the
    oracle's own bf16 emulation is ~19 % RMS away from its fp32 result at 4 x 256^2, ~50 % at 2 x 128^2), so this is a
    sanity bound - a layout or indexing bug gives uncorrelated outputs (RMS error sqrt(2)) - not a precision claim:
    the HIP result is no farther from the emulation than twice the emulation's own distance from fp32, and the gradient
    still points the fp32 way."""
    B, S = 4, 256
    spec, st = make_state(2, 8, 32, "audio")
    x = synth_inputs(B, S, seed=9)["audio"]
    masks = {b.idx: torch.ones(B) * (1.0 - b.drop_rate) for b in spec.blocks if b.skip}
    so = grad_state(st)
    (c, r, a), f = O.forward(so, x, 2, True, masks)
    loss = c.sum() * 0.01 + (r ** 2).mean() + sum((u ** 2).mean() for u in f)
    loss.backward()
    # the oracle's emulation of the same mode, forward AND backward: its distance from the fp32 oracle is the yardstick
    sb = grad_state(st)

    def emu():
        (cb_, rb_, _), fb_ = O.forward(sb, x, 2, True, masks)
        (cb_.sum() * 0.01 + (rb_ ** 2).mean() + sum((u ** 2).mean() for u in fb_)).backward()
        return (cb_.detach(), rb_.detach(), None), [u.detach() for u in fb_]
    (cb, rb, _), fb = _bf16_oracle(emu)
    net = Net(spec, DEV, trainable=True, precision=precision)
    net.load_state(st)
    skip = [b for b in spec.blocks if b.skip]
    ds = torch.ones(len(skip), B, device=DEV)
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=True, drop_scale=ds)
    noise = l2rel(rb, r)
    assert 1e-3 < noise < 0.5, noise
    assert l2rel(reg, rb) < 2 * noise and l2rel(reg, r) < 2 * noise, (l2rel(reg, rb), l2rel(reg, r), noise)
    for u, v, w in zip(feats, f, fb):
        assert l2rel(feat_nchw(u), w) < 2 * l2rel(w, v) + 1e-3
    dcls = (0.01 * cls * (1 - cls)).contiguous()
    dreg = (2.0 * reg / reg.numel()).contiguous()
    dfe = [(2.0 * u.z / u.z.numel()).contiguous() for u in feats]
    net.ps.grad.zero_()
    net.backward(dcls, dreg, dfe)
    torch.cuda.synchronize()
    grads = net.ps.export_grads()
    dot = n1 = n2 = 0.0
    for k, v in so.items():
        if not v.requires_grad:
            continue
        ref, got = v.grad.double(), grads[k].double()
        assert torch.isfinite(got).all()
        dot += float((ref * got).sum()); n1 += float((ref * ref).sum()); n2 += float((got * got).sum())
    cos, ratio = dot / (n1 ** 0.5 * n2 ** 0.5), (n2 / n1) ** 0.5

    def cosine(ga, gb):
        d = a2 = b2 = 0.0
        for k, v in so.items():
            if v.requires_grad and ga.get(k) is not None and gb.get(k) is not None:
                u, w = ga[k].double(), gb[k].double()
                d += float((u * w).sum()); a2 += float((u * u).sum()); b2 += float((w * w).sum())
        return d / (a2 ** 0.5 * b2 ** 0.5)

    g32 = {k: v.grad for k, v in so.items() if v.requires_grad}
    gem = {k: v.grad for k, v in sb.items() if v.requires_grad}
    cos_emu = cosine(gem, g32)                 # how far the RULE itself moves the gradient
    cos_hip_emu = cosine(grads, gem)
    print("D2 train %s: gradient cos HIP vs fp32 %.4f, emulation vs fp32 %.4f, HIP vs emulation %.4f, norm ratio %.3f" % (
        precision, cos, cos_emu, cos_hip_emu, ratio))
    # bound derived from the oracle's own emulation: the HIP gradient's angle to the fp32 gradient and to the emulation's is within
    # twice the emulation's own angle to fp32 (1 - cos is the squared-angle scale)
    assert (1 - cos) <= 2.0 * (1 - cos_emu) + 2e-2 and (1 - cos_hip_emu) <= 2.0 * (1 - cos_emu) + 2e-2, (cos, cos_emu, cos_hip_emu)
    assert 0.6 < ratio < 1.6, ratio


def _rows(t):      # NCHW -> NHWC rows [B*H*W, C] on the device
    return t.permute(0, 2, 3, 1).contiguous().view(-1, t.shape[1]).to(DEV)


@pytest.mark.parametrize("precision", ["fp32", "bf16"])
@pytest.mark.parametrize("case", ["d4_768_eval", "d2_256_train"])
def test_teacher_forced_units(case, precision):
    """The tight mid-level statement about the reduced-precision modes: every materialised unit of the HIP net -
    stem, each MBConv block, each BiFPN down-channel conv and node - is handed the INPUTS the oracle's emulation of the mode saw
    (engine `force_out`: the unit's own output is recorded, then overwritten with the emulation's, so rounding differences cannot compound
    across units) and its output is compared with the emulation's at 1e-2 of the tensor's largest value; the head outputs follow from the
    forced pyramid.  A mis-wired mode (wrong tensor rounded, wrong layout, a stale coefficient) shows up in the first unit it touches
    instead of drowning in the 0.4 cosine of a free-running net.  Frozen D4 at 768^2 (config 5's shapes, eval-mode BatchNorm) and the
    trainable D2's train-mode forward (live batch statistics, drop-connect masks); fp32 runs the same harness at 1e-4."""
    if case == "d4_768_eval":
        coef, cin, B, S, train, mod = 4, 3, 1, 768, False, "rgb"
    else:
        coef, cin, B, S, train, mod = 2, 8, 4, 256, True, "audio"
    spec, st = make_state(coef, cin, 31 if not train else 32, mod)
    x = synth_inputs(B, S, seed=8)[mod]
    masks = {b.idx: torch.ones(B) * (1.0 - b.drop_rate) for b in spec.blocks if b.skip} if train else None
    tap = {}

    def run():
        O.TAP = tap
        try:
            with torch.no_grad():
                return O.forward({k: v.clone() for k, v in st.items()}, x, coef, train, masks)
        finally:
            O.TAP = None
    (c, r, _), f = _bf16_oracle(run) if precision != "fp32" else run()
    net = Net(spec, DEV, trainable=train, precision=precision)
    net.load_state(st)
    net.probe, net.force_out = {}, {k: _rows(v) for k, v in tap.items()}
    net.begin_step()
    ds = None
    if train:      # masks of ones * keep -> scale 1 for every sample (same as the oracle call)
        ds = torch.ones(sum(1 for b in spec.blocks if b.skip), B, device=DEV)
    cls, reg, feats = net.forward(x.to(DEV), train=train, drop_scale=ds)
    torch.cuda.synchronize()
    # (train mode: BatchNorm over the batch's few samples - 64 per channel on the 4 x 4 level here - amplifies a unit's rounding differences)
    tol = 1e-4 if precision == "fp32" else (3e-2 if train else 1e-2)
    assert set(net.probe) == set(tap), (sorted(set(tap) ^ set(net.probe)))
    worst = []
    for name, ref in tap.items():
        got = net.probe[name]
        e = relerr(got, _rows(ref))
        worst.append((e, name))
        assert e <= tol, (name, e)
    worst.sort(reverse=True)
    print("%s %s: %d units, worst max-error / max-value %.2e (%s), median %.2e" % (case, precision, len(worst), worst[0][0], worst[0][1],
                                                                                   worst[len(worst) // 2][0]))
    # heads on the forced pyramid (the last cell's node outputs were overwritten with the emulation's).  Inside a head nothing is forced:
    # four sepconv + BatchNorm + swish layers and the header run free, and the random-weight classifier's sigmoid saturates - hence the
    # wider bound on the probabilities in the rounding modes (measured 8e-2 at D4; regression 6e-4)
    print("   heads on the forced pyramid: reg %.2e cls %.2e" % (relerr(reg, r), relerr(cls, c)))
    assert relerr(reg, r) <= tol and relerr(cls, c) <= (3 * tol if precision == "fp32" else 0.2), (relerr(reg, r), relerr(cls, c))


def test_teacher_pack_matches_separate_nets():
    """Round 4, grouped frozen nets: the three teachers (RGB, depth: 3 input channels; thermal: 1) evaluated as ONE batch of 3 x B images -
    one launch per layer, every workgroup reading its own net's weights (mmd_set_group, engine.pack_nets / Net.forward(pack=...)) - against
    the three nets evaluated one by one: class probabilities, regressions and the five pyramid maps of every net.  Same kernels and
    arithmetic; the tile shapes a launch picks depend on its row count, so sums are re-associated (measured 2e-5 of a tensor's largest value; bound 1e-4)."""
    from mm_distillnet_amd.engine import pack_nets
    B, S = 8, 512
    mods = {"rgb": (3, 11), "depth": (3, 14), "thermal": (1, 12)}
    slot = 32 * 28
    nets, outs, xs = [], [], []
    for m, (cin, seed) in mods.items():
        spec, st = make_state(2, cin, seed, m if m != "depth" else "rgb")
        net = Net(spec, DEV, trainable=False, stem_slot=slot)
        net.load_state(st)
        x = synth_inputs(B, S, seed=50 + seed)[m if m != "depth" else "rgb"].to(DEV)
        net.begin_step()
        c, r, f = net.forward(x, train=False)
        outs.append((c.clone(), r.clone(), [feat_nchw(u).clone() for u in f]))
        nets.append(net); xs.append(x)
    assert pack_nets(nets)
    for n in nets:
        n.refresh()
    nets[0].begin_step()
    c, r, f = nets[0].forward(xs, train=False, pack=nets)
    torch.cuda.synchronize()
    assert c.shape[0] == 3 * B and f[0].B == 3 * B
    for gi, (c1, r1, f1) in enumerate(outs):
        sl = slice(gi * B, (gi + 1) * B)
        assert relerr(c[sl], c1) < 1e-4 and relerr(r[sl], r1) < 1e-4, (gi, relerr(c[sl], c1), relerr(r[sl], r1))
        for u, v in zip(f, f1):
            assert relerr(feat_nchw(u)[sl], v) < 1e-4, (gi, relerr(feat_nchw(u)[sl], v))
    # the nets still work one by one after packing (their stores are views of the shared buffers now)
    nets[2].begin_step()
    c2, r2, _ = nets[2].forward(xs[2], train=False)
    assert relerr(c2, outs[2][0]) < 1e-4 and relerr(r2, outs[2][1]) < 1e-4      # (run to run: the squeeze-excite pools are fp32 atomics)
    # B = 2: the 4 x 4 level holds 32 rows per net - not whole 128-row tiles, but whole 32-row tiles: since round 5 the heads run such a
    # trailing level as plain launches on the skinny kernel, and the pack equals the nets one by one here too
    nets[0].begin_step()
    c, r, f = nets[0].forward([x[:2] for x in xs], train=False, pack=nets)
    c, r = c.clone(), r.clone()
    for gi in range(3):
        nets[gi].begin_step()
        c1, r1, _ = nets[gi].forward(xs[gi][:2], train=False)
        assert relerr(c[2 * gi:2 * gi + 2], c1) < 1e-4 and relerr(r[2 * gi:2 * gi + 2], r1) < 1e-4, gi
    # a geometry whose smallest level is not whole 32-row tiles per net (B = 1: 16 rows) is refused (the engine then keeps one net per stream)
    nets[0].begin_step()
    with pytest.raises(RuntimeError):
        nets[0].forward([x[:1] for x in xs], train=False, pack=nets)



def _pack_case(coef, B, S, fuse_node=True, raw_logits=False):
    from mm_distillnet_amd.engine import pack_nets
    mods = {"rgb": (3, 11), "depth": (3, 14), "thermal": (1, 12)}
    slot = 32 * 28 if coef == 2 else None
    nets, outs, xs = [], [], []
    for m, (cin, seed) in mods.items():
        spec, st = make_state(coef, cin, seed, m if m != "depth" else "rgb")
        if slot is None:
            slot = spec.stem_out * 28
        net = Net(spec, DEV, trainable=False, stem_slot=slot)
        net.FUSE_NODE = fuse_node
        net.load_state(st)
        x = synth_inputs(B, S, seed=50 + seed)[m if m != "depth" else "rgb"].to(DEV)
        net.begin_step()
        c, r, f = net.forward(x, train=False, raw_logits=raw_logits)
        outs.append((c.clone(), r.clone(), [feat_nchw(u).clone() for u in f]))
        nets.append(net); xs.append(x)
    assert pack_nets(nets)
    for n in nets:
        n.refresh()
    return nets, xs, outs


@pytest.mark.parametrize("coef,fuse_node", [(1, True), (2, False), (4, True)])
def test_teacher_pack_unfused_nodes(coef, fuse_node):
    """a pack forward whose BiFPN nodes take the two-launch path - an fpn width without a whole-node kernel (D1: 88) or
    MMD_NO_NODE_FUSE - must read every net's own fusion weights and depthwise taps (mmd_bifpn_node_dw_fwd honours the group descriptor
    now), not net 0's: each net's slice of the packed result equals that net evaluated alone."""
    # (4: BASELINE configs[4]'s geometry, D4 at 768^2 - the 6 x 6 level holds 288 rows per net, not whole 128-row tiles: the heads run that
    # level as plain launches on the 32-row skinny kernel, round 5; its large BiFPN levels take the two-launch node path)
    B, S = 8, (768 if coef == 4 else 512)
    # (D4's random-weight classifier saturates - logits of +-30, where a re-associated sum of another tile shape moves a probability by
    # up to 1.5e-3, measured, with the regression and the maps at 1e-5: the class head is compared on its LOGITS there, bound 5e-4 of
    # the largest)
    raw = coef == 4
    nets, xs, outs = _pack_case(coef, B, S, fuse_node, raw_logits=raw)
    assert fuse_node is False or not nets[0]._node_fusable(Feat_like(nets[0].spec.fpn_w))
    nets[0].begin_step()
    c, r, f = nets[0].forward(xs, train=False, pack=nets, raw_logits=raw)
    torch.cuda.synchronize()
    ctol = 5e-4 if coef == 4 else 1e-4
    for gi, (c1, r1, f1) in enumerate(outs):
        sl = slice(gi * B, (gi + 1) * B)
        assert relerr(c[sl], c1) < ctol and relerr(r[sl], r1) < 1e-4, (gi, relerr(c[sl], c1), relerr(r[sl], r1))
        for u, v in zip(f, f1):
            assert relerr(feat_nchw(u)[sl], v) < 1e-4, (gi, relerr(feat_nchw(u)[sl], v))


class Feat_like:
    def __init__(self, C):
        self.C, self.M = C, 1 << 20


def test_pack_refuses_group_unaware_launches():
    """inside a pack forward a launch either honours the group descriptor or is refused: a plain call() of a parameter-reading entry point
    raises on the host, and an entry point without a group mode issued through Net._c is reported by the library (mmd_set_group's clearing
    call returns -22 when nothing read the descriptor)."""
    import mm_distillnet_amd.engine as E
    spec, st = make_state(2, 3, 11, "rgb")
    net = Net(spec, DEV, trainable=False)
    net.load_state(st)
    x = torch.zeros(8 * 16, 16, device=DEV)
    y = torch.empty(8 * 4, 16, device=DEV)
    net._grp = (2, 4, net.ps.n_params, net.ps.bn_total)
    try:
        with pytest.raises(RuntimeError, match="does not honour the group descriptor"):
            net._c("mmd_maxpool_same_fwd", x, y, 8, 4, 4, 16)
        E._PACK.on = True
        with pytest.raises(RuntimeError, match="inside a pack forward"):
            E.call("mmd_affine_act", x, None, None, None, None, None, None, 0, None, 0, None, x, 128, 16)
        E.call("mmd_maxpool_same_fwd", x, y, 8, 4, 4, 16)       # parameter-free: allowed
    finally:
        E._PACK.on = False
        net._grp = None
    # the descriptor is thread-local: a launch from another host thread while this one holds a group runs ungrouped
    import threading
    dll = E._lib.LIB.load()
    assert dll.mmd_set_group(3, 8, 1024, 64) == 0
    res = {}

    def other():
        res["rc_clear"] = dll.mmd_set_group(1, 0, 0, 0)          # nothing set on THIS thread: plain OK

    t = threading.Thread(target=other); t.start(); t.join()
    assert res["rc_clear"] == 0
    assert dll.mmd_set_group(1, 0, 0, 0) == -22                  # this thread's group was never read by a launch
    torch.cuda.synchronize()


def test_frozen_nets_bit_reproducible():
    """the frozen nets' squeeze-excite pool sums are integer (Q36 fixed-point) atomics and the fused expand + depthwise
    kernel adds its waves' partials in a fixed order, so two evaluations of the teacher pack - and of a single net - on the same input give
    bit-identical class, regression and feature tensors (fp32 atomics made last bits, and now and then a pseudo-label, differ run to run)."""
    B, S = 8, 512
    nets, xs, outs = _pack_case(2, B, S)
    runs = []
    for _ in range(3):
        nets[0].begin_step()
        c, r, f = nets[0].forward(xs, train=False, pack=nets)
        torch.cuda.synchronize()
        runs.append((c.clone(), r.clone(), [u.z.clone() for u in f]))
    for c, r, f in runs[1:]:
        assert torch.equal(c, runs[0][0]) and torch.equal(r, runs[0][1])
        assert all(torch.equal(a, b) for a, b in zip(f, runs[0][2]))
    # one net alone, twice (different tile shapes than the pack: compared with itself)
    single = []
    for _ in range(2):
        nets[2].begin_step()
        c, r, f = nets[2].forward(xs[2], train=False)
        torch.cuda.synchronize()
        single.append((c.clone(), r.clone(), [u.z.clone() for u in f]))
    assert torch.equal(single[0][0], single[1][0]) and torch.equal(single[0][1], single[1][1])
    assert all(torch.equal(a, b) for a, b in zip(single[0][2], single[1][2]))


# ---- D4 (BASELINE configs[4]) against fixtures made by the reference's own D4 classes (tools/oracle/make_golden.py golden_net_d4)
@pytest.mark.parametrize("mod,cin,seed", [("rgb", 3, 41), ("thermal", 1, 42), ("audio", 8, 43)])
def test_net_d4_768_eval_golden(golden_dir, mod, cin, seed):
    gold = np.load(os.path.join(golden_dir, f"net_d4_768_eval_{mod}.npz"))
    spec, st = make_state(4, cin, seed, mod)
    net = Net(spec, DEV, trainable=False)
    net.load_state(st)
    x = synth_inputs(1, 768, seed=44)[mod]
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=False)
    torch.cuda.synchronize()
    assert tuple(cls.shape) == (1, 110484, 20)
    a = net.anchors(768).cpu()
    np.testing.assert_array_equal(a[::997].numpy(), gold["anchors.sample"])
    check_summary(gold, "anchors", a, 1e-6, 1e-7)
    check_summary(gold, "cls", cls, 1e-3, 1e-4); check_summary(gold, "reg", reg, 1e-3, 1e-4)
    for i, u in enumerate(feats):
        check_summary(gold, f"feat{i}", feat_nchw(u).contiguous(), 1e-3, 1e-4)


def test_net_d4_train_fwd_bwd_golden(golden_dir):
    gold = np.load(os.path.join(golden_dir, "net_d4_256_train_audio.npz"))
    spec, st = make_state(4, 8, 43, "audio")
    x = synth_inputs(2, 256, seed=45)["audio"]
    masks = {int(b): torch.from_numpy(m) for b, m in zip(gold["drop_blocks"], gold["drop_masks"])}
    net = Net(spec, DEV, trainable=True)
    net.load_state(st)
    skip = [b for b in spec.blocks if b.skip]
    ds = torch.stack([masks[b.idx] / (1.0 - b.drop_rate) for b in skip]).to(DEV)
    net.begin_step()
    cls, reg, feats = net.forward(x.to(DEV), train=True, drop_scale=ds)
    check_summary(gold, "cls", cls, 1e-3, 1e-4); check_summary(gold, "reg", reg, 1e-3, 1e-4)
    for i, u in enumerate(feats):
        check_summary(gold, f"feat{i}", feat_nchw(u).contiguous(), 1e-3, 1e-4)
    loss = cls.sum() * 0.01 + (reg ** 2).mean() + sum((u.z ** 2).mean() for u in feats)
    assert abs(loss.item() - float(gold["loss"])) < 1e-3 * abs(float(gold["loss"]))
    dcls = (0.01 * cls * (1 - cls)).contiguous()
    dreg = (2.0 * reg / reg.numel()).contiguous()
    dfe = [(2.0 * u.z / u.z.numel()).contiguous() for u in feats]
    net.ps.grad.zero_()
    net.backward(dcls, dreg, dfe)
    torch.cuda.synchronize()
    grads = net.ps.export_grads()
    ex = net.ps.export_state()
    for k in gold.files:
        if k.startswith("stat.") and k.endswith(".head"):
            name = k[5:-5]
            check_summary(gold, "stat." + name, ex[name], 1e-4, 1e-5)
        if k.startswith("gradnorm."):
            top = k[len("gradnorm."):]
            tot = sum(float(grads[n].double().pow(2).sum()) for n in grads
                      if (".".join(n.split(".")[:2]) if n.startswith("bifpn") else n.split(".")[0]) == top)
            assert abs(tot ** 0.5 - float(gold[k])) <= 1e-2 * float(gold[k]), (top, tot ** 0.5, float(gold[k]))
        if k.startswith("grad.") and k.endswith(".head"):
            name = k[5:-5]
            # (D4: 32 MBConv blocks + 7 BiFPN cells deep, BatchNorm over 8 samples on the 2 x 2 level: the trainable net's f64 / fp32 atomics
            # (BatchNorm sums, depthwise weight gradients) arrive in another order every run, and this chain amplifies that to 2e-3 .. 6.5e-3
            # of a watched tensor's largest element - measured over this round's runs, the stem and the last block's BatchNorm weight worst;
            # D2 at 128^2 holds 1e-3.  The per-module gradient NORMS above hold 1e-2 in every run.)
            check_summary(gold, "grad." + name, grads[name], 2e-2, 1e-2)
    assert int(ex["backbone_net.model._bn0.num_batches_tracked"]) == int(gold["nbt"])
