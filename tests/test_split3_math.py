"""CPU statement of the arithmetic behind the "split" GEMM form (csrc/common.h, mmd_split3_pk): an fp32 value splits EXACTLY into three bf16
pieces by round-to-nearest, bf16 x bf16 products are exact in fp32, and the three partial products the kernels drop amount to about one
fp32 rounding (bounded by 2^-23 of the product, 2^-27 rms), while the form saves ten of every sixteen accumulator roundings.  torch's float32 -> bfloat16 conversion is the same round-to-nearest-even the kernels use (v_cvt_pk_bf16_f32).  No GPU."""
import numpy as np
import torch


def split3(x: torch.Tensor):
    h = x.to(torch.bfloat16).to(torch.float32)
    r = x - h
    m = r.to(torch.bfloat16).to(torch.float32)
    s = r - m
    l = s.to(torch.bfloat16).to(torch.float32)
    return h, m, l, r, s


def _values():
    g = torch.Generator().manual_seed(7)
    v = [torch.randn(200000, generator=g),
         torch.randn(200000, generator=g) * torch.exp2(torch.randint(-100, 100, (200000,), generator=g).float()),
         torch.tensor([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 1.0 - 2.0 ** -24, 3.0, 2.0 ** -120, -(2.0 ** 100), 1.9999999, 0.1, 1e-30, 65504.0,
                       float(np.float32(np.pi)), 2.0 ** 127]),
         # random bit patterns with exponents 2^-100 .. 2^100 (below ~2^-110 the third piece, 2^-16 of the value, falls into bf16's subnormal
         # range and loses bits: an ABSOLUTE error below 2^-133, outside anything a network holds)
         torch.from_numpy(np.random.default_rng(3).integers((127 - 100) << 23, (127 + 100) << 23, 200000, dtype=np.int64).astype(np.uint32).view(np.float32).copy())]
    return torch.cat(v)


def test_three_way_split_is_exact():
    x = _values()
    h, m, l, r, s = split3(x)
    x64 = x.double()
    # both residuals are exact in fp32 (so computing them in fp32, as the kernels do, loses nothing) ...
    assert torch.equal(r.double(), x64 - h.double())
    assert torch.equal(s.double(), x64 - h.double() - m.double())
    # ... and the third piece takes what is left: h + m + l == x, exactly
    assert torch.equal(h.double() + m.double() + l.double(), x64)
    # piece sizes: |m| <= 2^-8 |x|, |l| <= 2^-16 |x| (half an ulp of an 8-bit significand at each level, with the round-up slack)
    nz = x != 0
    assert float((m[nz].abs() / x[nz].abs()).max()) <= 2.0 ** -8
    assert float((l[nz].abs() / x[nz].abs()).max()) <= 2.0 ** -16


def test_bf16_products_are_exact_in_fp32_and_the_dropped_terms_are_small():
    g = torch.Generator().manual_seed(11)
    a = torch.randn(300000, generator=g) * torch.exp2(torch.randint(-20, 20, (300000,), generator=g).float())
    b = torch.randn(300000, generator=g) * torch.exp2(torch.randint(-20, 20, (300000,), generator=g).float())
    ah, am, al, _, _ = split3(a)
    bh, bm, bl, _, _ = split3(b)
    # a bf16 x bf16 product has a 16-bit significand: the fp32 product IS the exact product
    for p, q in ((ah, bh), (ah, bm), (am, bh), (ah, bl), (al, bh), (am, bm)):
        assert torch.equal((p * q).double(), p.double() * q.double())
    kept = (ah.double() * bh.double() + ah.double() * bm.double() + am.double() * bh.double()
            + ah.double() * bl.double() + al.double() * bh.double() + am.double() * bm.double())
    exact = a.double() * b.double()
    rel = (kept - exact).abs() / exact.abs()          # the three dropped partial products: am*bl + al*bm + al*bl
    # |m| <= 2^-8 |x| and |l| <= 2^-16 |x|: bounded by 2^-23 (+ 2^-32); on random operands at most about 2^-24, 2^-27 rms
    assert rel.max().item() <= 2.0 ** -23 and rel.max().item() <= 2.0 ** -24 * 1.2 and rel.pow(2).mean().sqrt().item() <= 2.0 ** -27
    # for scale: rounding the exact product to fp32 ONCE costs up to 2^-24, 2^-25.3 rms - four times the dropped terms' rms
    one = (exact.float().double() - exact).abs() / exact.abs()
    assert one.max().item() > 2.0 ** -24.2 and one.pow(2).mean().sqrt().item() > 4 * rel.pow(2).mean().sqrt().item()


def test_split_dot_product_is_as_accurate_as_an_fp32_fma_chain():
    """A K = 256 dot product: six-term split products accumulated in fp32 per 16-deep group (what the MFMA does, modelled with exact group sums
    rounded once) against a sequential fp32 fma chain, both against float64."""
    g = torch.Generator().manual_seed(5)
    n, K = 4000, 256
    a = torch.randn(n, K, generator=g); b = torch.randn(n, K, generator=g)
    ref = (a.double() * b.double()).sum(1)
    mag = (a.double() * b.double()).abs().sum(1)
    acc = torch.zeros(n)
    for k in range(K):                                 # fma chain: one rounding per term
        acc = (acc.double() + a[:, k].double() * b[:, k].double()).float()
    e_chain = ((acc.double() - ref).abs() / mag)
    ah, am, al, _, _ = split3(a); bh, bm, bl, _, _ = split3(b)
    acc = torch.zeros(n)
    for g0 in range(0, K, 16):
        sl = slice(g0, g0 + 16)
        for p, q in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):      # smallest partial products first, as the kernels issue them
            acc = (acc.double() + (p[:, sl].double() * q[:, sl].double()).sum(1)).float()
    e_split = ((acc.double() - ref).abs() / mag)
    assert e_split.max().item() <= 1.25 * e_chain.max().item() and e_split.pow(2).mean().sqrt().item() <= e_chain.pow(2).mean().sqrt().item()
