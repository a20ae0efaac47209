"""NumPy emulation that chose the tolerance-mode encoder's arithmetic (csrc/gemm_split_f16.hip): the five-layer SDAV chain
with every operand split into 16-bit pieces and a layer computed as a few products of pieces, against the fp64 oracle on real
and random frames, N(0,1) and 1/sqrt(fan_in) weights.  Run here (CPU, minutes):  python scripts/emul_split_encoder.py [bf16]
  fp16 x 2 pieces, 3 products, fp32 accumulate: relative L2 max 1.7e-5 (N(0,1)) / 7e-8 (fan_in)   <- built
  bf16 x 3 pieces: 3 products 2.7e-4, 4 products 2.3e-4 (both over north_star's 1e-4), 6 products 1e-5
Test infrastructure only (imports oracle/)."""
import sys
if len(sys.argv) > 1 and sys.argv[1] == 'bf16':
    import sys, numpy as np
    sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
    import config1_common as c1
    from oracle import sdav as osdav
    def bf16(x):  # round-to-nearest-even to bf16, returned as float32
        x = np.asarray(x, dtype=np.float32)
        u = x.view(np.uint32).astype(np.uint64)
        r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
        return r.astype(np.uint32).view(np.float32)
    def split(x, n):
        x = np.asarray(x, dtype=np.float64)
        out = []
        r = x.copy()
        for _ in range(n):
            p = bf16(r.astype(np.float32)).astype(np.float64)
            out.append(p); r = r - p
        return out
    def layer(h, w, b, nprod, acc32):
        hs, ws = split(h, 3), split(w, 3)
        pairs = {3: [(0,0),(0,1),(1,0)], 4: [(0,0),(0,1),(1,0),(1,1)], 6: [(0,0),(0,1),(1,0),(0,2),(1,1),(2,0)]}[nprod]
        z = np.zeros((h.shape[0], w.shape[1]))
        for i, j in pairs:
            if acc32:
                z += (hs[i].astype(np.float32) @ ws[j].astype(np.float32)).astype(np.float64)
            else:
                z += hs[i] @ ws[j]
        return 1.0 / (1.0 + np.exp(-(z + b)))
    x_real = c1.oracle_patches(c1.frame_paths())[:10]
    x_rand = np.random.RandomState(0).uniform(0, 1, size=(10, 30, 1681))
    for scale in ("reference", "fan_in"):
        ws, bs = osdav.init_weights(4, scale=scale)
        for name, x in (("real", x_real), ("rand", x_rand)):
            ref = osdav.transform(x, ws, bs)
            for nprod in (3, 4, 6):
                for acc32 in (False, True):
                    h = x.reshape(-1, 1681)
                    for w, b in zip(ws, bs):
                        h = layer(h, w, b, nprod, acc32)
                    l2 = np.linalg.norm(h - ref, axis=1) / np.linalg.norm(ref, axis=1)
                    print(scale, name, "products", nprod, "acc32" if acc32 else "acc64", "rel L2 max %.3g median %.3g  max abs %.3g" % (l2.max(), np.median(l2), np.abs(h - ref).max()), flush=True)
else:
    import sys, numpy as np
    sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
    import config1_common as c1
    from oracle import sdav as osdav
    def split16(x, n, scale):
        x = np.asarray(x, dtype=np.float64) * scale
        out = []; r = x.copy()
        for _ in range(n):
            p = r.astype(np.float16).astype(np.float64)
            out.append(p); r = r - p
        return out
    def layer(h, w, b, sx, sw, one_acc):
        hs, ws = split16(h, 2, sx), split16(w, 2, sw)
        pairs = [(0,0),(0,1),(1,0)]
        if one_acc:   # one fp32 accumulator over all three products, K in chunks of 32 (MFMA steps) -- emulate by float32 sum of chunked partials
            K = h.shape[1]
            z = np.zeros((h.shape[0], w.shape[1]), dtype=np.float32)
            for k0 in range(0, K, 256):
                for i, j in pairs:
                    z = (z + (hs[i][:, k0:k0+256].astype(np.float32) @ ws[j][k0:k0+256].astype(np.float32))).astype(np.float32)
            z = z.astype(np.float64)
        else:
            z = sum(hs[i] @ ws[j] for i, j in pairs)
        z = z / (sx * sw)
        return 1.0 / (1.0 + np.exp(-(z + b)))
    x_real = c1.oracle_patches(c1.frame_paths())[:10]
    x_rand = np.random.RandomState(0).uniform(0, 1, size=(10, 30, 1681))
    for scale in ("reference", "fan_in"):
        ws, bs = osdav.init_weights(4, scale=scale)
        for name, x in (("real", x_real), ("rand", x_rand)):
            ref = osdav.transform(x, ws, bs)
            for one_acc in (False, True):
                h = x.reshape(-1, 1681)
                for w, b in zip(ws, bs):
                    sw = 2.0 ** np.floor(np.log2(4096.0 / np.abs(w).max()))
                    h = layer(h, w, b, 2048.0, sw, one_acc)
                l2 = np.linalg.norm(h - ref, axis=1) / np.linalg.norm(ref, axis=1)
                print(scale, name, "f16x2 3 products", "fp32 acc" if one_acc else "exact acc", "rel L2 max %.3g median %.3g  max abs %.3g" % (l2.max(), np.median(l2), np.abs(h - ref).max()), flush=True)
