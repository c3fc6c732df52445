"""Load-time weight layouts (torch ops used as plumbing: permute / pad / cast, no arithmetic on the hot path).

* `pack_frag`      — [N][K] row-major -> MFMA fragment order [N/16][K/32][64 lanes][8] consumed by the decode GEMMs
                      (csrc/gemm_skinny.hip): lane l = g*16 + r holds row r of the 16-row tile, k = 8g..8g+7.
* `conv_weight`    — torch Conv1d weight [Cout][Cin][k] -> [Cout][k][Cin_pad32] (k-major so that a tap is a row shift
                      of the time-major activation; csrc/gemm_tiled.hip).
* `fold_weight_norm` — w = v * (g / ||v||), the new-style parametrisation of hifigan/generator.py:26-29.
"""
import torch


def pad_to(x, dim, mult):
    n = x.shape[dim]
    tgt = (n + mult - 1) // mult * mult
    if tgt == n:
        return x
    shape = list(x.shape)
    shape[dim] = tgt - n
    return torch.cat([x, x.new_zeros(shape)], dim=dim)


def pack_frag(w):
    """[N][K] -> flat tensor in [N/16][K/32][g=4][r=16][8] order (N padded to 16, K must be a multiple of 32)."""
    assert w.dim() == 2 and w.shape[1] % 32 == 0, w.shape
    w = pad_to(w, 0, 16)
    N, K = w.shape
    return w.view(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).contiguous().view(N, K)


def pack_narrow4(w):
    """[N][K] -> flat tensor in [N/4][K/128][g=4][ksub=4][c=4][8] order (N % 4 == 0, K % 128 == 0): lane (c + 4*ksub, g) of the
    4-column workgroup form (csrc/gemm_skinny.hip: gemm_narrow_resid_kernel) holds W[4*nt + c][128*kb + 32*ksub + 8*g ..+8]."""
    assert w.dim() == 2 and w.shape[0] % 4 == 0 and w.shape[1] % 128 == 0, w.shape
    N, K = w.shape
    return w.view(N // 4, 4, K // 128, 4, 4, 8).permute(0, 2, 4, 3, 1, 5).contiguous().view(N, K)


def qkv_row_perm(n_heads, head_dim=64):
    """Row order of the fused [q | k | v] projection consumed by the SK_QKV_ROPE epilogue: inside every 64-row head, 16-row tile t
    holds d = 8t..8t+7 followed by their rotate-half partners d + 32, so a RoPE pair sits in one MFMA tile (lanes fr and fr ^ 8)."""
    assert head_dim == 64
    idx = []
    for h in range(n_heads):
        for t in range(4):
            idx += [h * 64 + 8 * t + i for i in range(8)] + [h * 64 + 32 + 8 * t + i for i in range(8)]
    return torch.tensor(idx, dtype=torch.long)


def pack_gate_up(wg, wu):
    """Interleave gate / up projections as alternating 16-row tiles, then pack (SwiGLU epilogue pairs tile 2p with 2p+1)."""
    I, H = wg.shape
    assert I % 16 == 0 and wu.shape == wg.shape
    g = wg.view(I // 16, 1, 16, H)
    u = wu.view(I // 16, 1, 16, H)
    return pack_frag(torch.cat([g, u], dim=1).reshape(2 * I, H))


def conv_weight(w, cin_pad=None):
    """[Cout][Cin][k] -> [Cout][k][Cin_pad] (flattened to [Cout][k*Cin_pad])."""
    cout, cin, k = w.shape
    cp = cin_pad or (cin + 31) // 32 * 32
    o = w.new_zeros(cout, k, cp)
    o[:, :, :cin] = w.permute(0, 2, 1)
    return o.reshape(cout, k * cp).contiguous()


def grouped_conv_weight(w, groups):
    """[D][Cg][k] (groups blocks of Cg output channels) -> [groups][Cg][k*Cg]."""
    D, cg, k = w.shape
    assert D % groups == 0 and D // groups == cg and cg % 32 == 0
    return w.view(groups, cg, cg, k).permute(0, 1, 3, 2).reshape(groups, cg, k * cg).contiguous()


def linear_weight(w, k_pad=None):
    """[N][K] with K zero-padded to a multiple of 32."""
    return pad_to(w, 1, 32).contiguous()


def fold_weight_norm(sd, name):
    k0 = name + '.parametrizations.weight.original0'
    if k0 in sd:
        g = sd[k0].float()
        v = sd[name + '.parametrizations.weight.original1'].float()
        return v * (g / v.norm(2, dim=(1, 2), keepdim=True))
    return sd[name + '.weight'].float()
