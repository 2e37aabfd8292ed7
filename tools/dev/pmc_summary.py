"""HBM traffic of ONE eager step (the dispatches after the last Adam launch of a `rocprofv3 --pmc X` run of bench.py),
per kernel family, with the gfx950 FETCH_SIZE correction (x2 for wide coalesced reads; MI355X_MICROARCH.md HBM section).
Usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv>"""
import collections, csv, sys

FAM = {"pw_gemm": ("pw_gemm_kernel", "pw_gemm_skinny_kernel", "pw_stream_kernel", "pw_rows_kernel", "pw_longk_kernel", "mbconv_expand_bwd_kernel", "pw_slab_kernel", "pw_slab_combine_kernel"),
       "pw_wgrad": ("pw_wgrad_kernel", "wgrad_grouped_kernel", "wgrad_fold_kernel", "wgrad_grouped_rect_kernel", "wgrad_fold_rect_kernel", "wgrad_grouped_split_kernel"),
       "dw_fwd": ("dw_fwd_kernel", "fuse_dw_fwd_kernel", "dw3_rows_kernel"), "dw_bwd": ("dw_wgrad_kernel", "dw_bwd_data_s2_kernel", "dw3_wgrad_rows_kernel"),
       "bn_bwd": ("bn_bwd_reduce_kernel", "bn_bwd_apply_kernel"), "mbx": ("mbx_kernel", "bifpn_node_fused_kernel"),
       "se": ("se_hidden_kernel", "se_gate_kernel", "se_bwd_a_kernel", "se_bwd_b_kernel", "se_bwd_ab_kernel"), "node_bwd": ("fuse_dw_bwd_kernel",)}


def final_eager_step(rows):
    """rows (sorted by dispatch) behind the last optimizer launch = bench.py's eager single-stream step(s).  Since round 6 the bench runs that
    step twice (a warm one, then the one its HIP events bracket).  A kernel that runs exactly once per step marks the period: the last
    `period` launches are the final step (the warm step may differ by one-off launches, so the halves are not compared)."""
    ad = [i for i, r in enumerate(rows) if "adam2_kernel" in r["Kernel_Name"]]
    seg = rows[ad[-1] + 1:]
    for marker in ("focal_finalize_kernel", "mta_kl_multi_kernel"):
        pos = [i for i, r in enumerate(seg) if marker in r["Kernel_Name"]]
        if len(pos) >= 2:
            return seg[len(seg) - (pos[-1] - pos[-2]):]
        if len(pos) == 1:
            return seg
    return seg


def last_step(path):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    seg = final_eager_step(rows)
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in seg:
        name = r["Kernel_Name"]
        for fam, keys in FAM.items():
            if any(k in name for k in keys):
                out[fam][0] += 1
                out[fam][1] += float(r["Counter_Value"]) * 1024.0      # counters are in KiB
    return out


f, w = last_step(sys.argv[1]), last_step(sys.argv[2])
print("family,launches,fetch_bytes_corrected(2x),write_bytes,total_hbm_bytes_per_step")
for fam in FAM:
    fb, wb = 2.0 * f[fam][1], w[fam][1]
    print("%s,%d,%.4g,%.4g,%.4g" % (fam, f[fam][0], fb, wb, fb + wb))


# per-variant view of the BiFPN node backward (where does its 2.3x traffic go?): template arguments = <MODE (bit 0 in1,
# bit 1 up, bit 2 pool), GEMM, NKK, CW>
def variants(path, key="fuse_dw_bwd_kernel"):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Dispatch_Id"]))
    out = collections.defaultdict(lambda: [0, 0.0])
    for r in final_eager_step(rows):
        n = r["Kernel_Name"]
        if key in n:
            k = n[n.index(key):].split("(")[0] + " grid " + r.get("Grid_Size", "?")
            out[k][0] += 1; out[k][1] += float(r["Counter_Value"]) * 1024.0
    return out


if len(sys.argv) > 3:
    fv, wv = variants(sys.argv[1]), variants(sys.argv[2])
    with open(sys.argv[3], "w") as fh:
        fh.write("BiFPN node backward, HBM bytes per launch by kernel variant (FETCH_SIZE x 2 + WRITE_SIZE, one eager step)\n")
        for k in sorted(fv, key=lambda k_: -(2 * fv[k_][1] + wv[k_][1])):
            n = max(fv[k][0], 1)
            fh.write("%-72s n=%2d  read %7.2f MB  written %7.2f MB  per launch\n" % (k, fv[k][0], 2 * fv[k][1] / n / 1e6, wv[k][1] / n / 1e6))
