"""Per-layer timing of the sparse-conv kernels on a synthetic S50k batch (dev tool, GPU only)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cagroup3d_amd import me, synthetic, _lib


def timeit(fn, iters=10, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def main():
    me.PRECISION = int(os.environ.get('PREC', '0'))
    bs = int(os.environ.get("BS", "4"))
    batch = synthetic.make_batch("S50k", bs)
    pts = torch.from_numpy(batch["points"]).cuda()
    coords = pts[:, :4].clone()
    coords[:, 1:] /= 0.02
    if os.environ.get("SORT"):
        c = coords.floor().long()
        if os.environ["SORT"] == "morton":
            def spread(v):
                v = v & 0x1FFFFF
                v = (v | (v << 32)) & 0x1F00000000FFFF
                v = (v | (v << 16)) & 0x1F0000FF0000FF
                v = (v | (v << 8)) & 0x100F00F00F00F00F
                v = (v | (v << 4)) & 0x10C30C30C30C30C3
                v = (v | (v << 2)) & 0x1249249249249249
                return v
            key = (c[:, 0] << 60) | (spread(c[:, 1] + 2048) << 2) | (spread(c[:, 2] + 2048) << 1) | spread(c[:, 3] + 2048)
        else:
            key = ((c[:, 0] * 4096 + c[:, 1] + 2048) * 4096 + c[:, 2] + 2048) * 4096 + c[:, 3] + 2048
        order = key.argsort()
        coords, pts = coords[order].contiguous(), pts[order].contiguous()
    t0 = time.time()
    x = me.SparseTensor(coordinates=coords, features=pts[:, 4:] / 255.)
    torch.cuda.synchronize()
    print("voxelise: N1=%d  (%.1f ms incl. first-call overheads)" % (len(x), (time.time() - t0) * 1e3))
    print("build map again: %.3f ms" % timeit(lambda: me._build_map(x.C, 1), 5, 1))
    mgr = x.coordinate_manager
    keys = {1: x.coordinate_map_key}
    for ts in (2, 4, 8, 16, 32):
        keys[ts] = mgr.stride(keys[ts // 2], 2)
    for ts in keys:
        print("ts=%d rows=%d" % (ts, mgr.get(keys[ts]).n))
    layers = [(1, 1, 3, 64, 3), (1, 1, 64, 64, 3), (1, 2, 64, 64, 3), (2, 2, 64, 64, 3), (2, 4, 64, 128, 3), (4, 4, 128, 128, 3),
              (4, 8, 128, 256, 3), (8, 8, 256, 256, 3), (8, 16, 256, 512, 3), (16, 16, 512, 512, 3), (16, 32, 512, 512, 3)]
    tot = 0
    for tin, tout, cin, cout, ks in layers:
        t0 = time.time()
        km = mgr.kernel_map(keys[tin], keys[tout], ks, 1, False)
        _ = km.nbrT
        torch.cuda.synchronize()
        tmap = (time.time() - t0) * 1e3
        P = int((km.nbr >= 0).sum())
        xin = torch.randn(km.n_in, cin, device="cuda")
        w = torch.randn(ks ** 3, cin, cout, device="cuda") * 0.05
        dy = torch.randn(km.n_out, cout, device="cuda")
        wt = w.transpose(1, 2).contiguous()
        tf_i = timeit(lambda: me._conv_fwd_raw(xin, w, km.nbr, None, km.n_out))
        t0 = time.time()
        pin, pout, off, P2 = km.pairs()
        seg, nseg = km.segments(128)
        torch.cuda.synchronize()
        tpairs = (time.time() - t0) * 1e3
        tf = timeit(lambda: me._conv_pairs(xin, w, pin, pout, seg, nseg, None, km.n_out))
        td = timeit(lambda: me._conv_pairs(dy, wt, pout, pin, seg, nseg, None, km.n_in))
        dw = torch.empty_like(w)
        lib = _lib.get()
        from ctypes import c_int32, c_int64
        def wg():
            lib.call("cg3d_spconv_wgrad", _lib.ptr(xin), _lib.ptr(dy), _lib.ptr(km.nbr), _lib.ptr(dw), c_int64(km.n_in),
                     c_int64(km.n_out), c_int32(ks ** 3), c_int32(cin), c_int32(cout), c_int32(0), lib.stream())
        tw_i = timeit(wg)
        wseg, nwseg = km.segments(me._wgrad_seg_len(P, cin, cout))
        def wgp():
            lib.call("cg3d_spconv_pairs_wgrad", _lib.ptr(xin), _lib.ptr(dy), _lib.ptr(pin), _lib.ptr(pout), _lib.ptr(wseg),
                     c_int64(nwseg), _lib.ptr(dw), c_int32(ks ** 3), c_int32(cin), c_int32(cout), c_int32(0), lib.stream())
        tw = timeit(wgp)
        ti_b = tw_b = float("nan")
        if me.PRECISION == 1 and cin % 8 == 0:
            wb = me._prep_bf16_t(w)
            rows16 = os.environ.get("ROWS16", "1") == "1"
            xg = me._to_bf16(xin) if rows16 else xin
            dg = me._to_bf16(dy) if (rows16 and cout % 8 == 0) else dy
            tcv = timeit(lambda: me._to_bf16(xin))
            ti_b = timeit(lambda: me._conv_implicit_bf16(xg, wb, km.nbr, None, km.n_out, cin, cout, P))
            tf = timeit(lambda: me._conv_pairs(xg, w, pin, pout, seg, nseg, None, km.n_out))
            bseg, nbseg = km.segments(me._wgrad_seg_len(P, cin, cout, 1))
            wp = 2 if (rows16 and dg is not dy) else 1
            xw, dw_ = (xg, dg) if wp == 2 else (xin, dy)
            def wgb():
                lib.call("cg3d_spconv_pairs_wgrad", _lib.ptr(xw), _lib.ptr(dw_), _lib.ptr(pin), _lib.ptr(pout), _lib.ptr(bseg),
                         c_int64(nbseg), _lib.ptr(dw), c_int32(ks ** 3), c_int32(cin), c_int32(cout), c_int32(wp), lib.stream())
            tw_b = timeit(wgb)
            print("   rows16=%s  to_bf16 %.3f ms" % (rows16, tcv))
        gf = 2.0 * P * cin * cout / 1e9
        gfd = 2.0 * km.n_out * ks ** 3 * cin * cout / 1e9
        tot += tf + td + tw
        print("ts %2d->%2d %4d->%4d k%d rows %7d pairs %8d (occ %.1f/%d) eff %7.2f GF dense %7.2f GF | map %5.1f+%4.1f ms | pairs: fwd %7.3f ms (%6.1f TF eff) dgrad %7.3f wgrad %7.3f | implicit: fwd %7.3f wgrad %7.3f | implicit bf16 fwd %7.3f bf16 wgrad %7.3f"
              % (tin, tout, cin, cout, ks, km.n_out, P, P / max(km.n_out, 1), ks ** 3, gf, gfd, tmap, tpairs, tf, gf / tf, td, tw, tf_i, tw_i, ti_b, tw_b))
    print("sum fwd+dgrad+wgrad of listed layers: %.2f ms" % tot)


if __name__ == "__main__":
    main()
