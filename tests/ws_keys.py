"""Rebuild the vertex-id -> edge-key map of a HIP extraction on the host from the workspace's sign
bitfield and vertex-id records (include/p3d_mc.h: p3d_mc_debug_layout), so meshes produced by the
one-pass kernel (whose vertex order is chunk arrival order) can be put in canonical form.  Also
checks the records themselves: every id in [0, V) must be assigned exactly once."""
import numpy as np


def vertex_keys_from_workspace(ws: np.ndarray, shape, nv: int, layout: dict, halo_last_plane: bool = False) -> np.ndarray:
    rx, ry, rz = shape
    ncz, U = layout["chunks_per_row"], layout["num_units"]
    assert U == rx * ry * ncz
    bits = ws[layout["off_bits"]:layout["off_bits"] + U * 8].view(np.uint64).reshape(rx, ry, ncz)
    rec = ws[layout["off_records"]:layout["off_records"] + U * 8].view(np.uint32).reshape(rx, ry, ncz, 2)
    inside = np.unpackbits(bits.view(np.uint8).reshape(rx, ry, ncz * 8), axis=-1, bitorder="little").astype(bool)
    assert not inside[:, :, rz:].any(), "sign bits beyond rz must be zero"
    zpad = ncz * 64
    valid = np.zeros((rx, ry, zpad), bool)
    valid[:, :, :rz] = True
    cx = np.zeros((rx, ry, zpad), bool)
    cy = np.zeros_like(cx)
    cz = np.zeros_like(cx)
    cx[:-1] = (inside[:-1] != inside[1:]) & valid[:-1]
    cy[:, :-1] = (inside[:, :-1] != inside[:, 1:]) & valid[:, :-1]
    cz[:, :, :rz - 1] = inside[:, :, :rz - 1] != inside[:, :, 1:rz]
    if halo_last_plane:  # in-plane edges of a halo plane belong to the next rank (include/p3d_mc.h: p3d_mc_slab)
        cy[-1] = False
        cz[-1] = False
    keys = np.full((nv,), -1, dtype=np.int64)
    lin = (np.arange(rx)[:, None, None] * ry + np.arange(ry)[None, :, None]) * rz + np.arange(zpad)[None, None, :]
    base = rec[..., 0].astype(np.int64)
    hdr = ws[:8192].view(np.uint64)
    if int(hdr[3]) == 1:  # H_RECFORM: the one-pass call left region * 2^26 + slot; dense base = slot + region prefix
        prefix = hdr[32 + 32 * 16:32 + 32 * 16 + 32].astype(np.int64)  # H_PREFIX
        region = np.minimum(base >> 26, 31)
        base = (base & 0x3FFFFFF) + prefix[region]
    if int(hdr[3]) == 2:  # region-layout form: rec[].x is a ROW; rows at or beyond V were moved into the holes below V
        H_LAYOUT, H_OCC, H_TS, H_TP, H_HS, H_HP, H_TE, NI = 600, 650, 700, 750, 800, 850, 900, 40
        assert int(hdr[0]) == nv
        first = hdr[H_LAYOUT:H_LAYOUT + NI + 1].astype(np.int64)       # 32 regions, 8 spill areas, the end
        occ = hdr[H_OCC:H_OCC + NI].astype(np.int64)
        ts, tp = hdr[H_TS:H_TS + NI].astype(np.int64), hdr[H_TP:H_TP + NI].astype(np.int64)
        hs, hp = hdr[H_HS:H_HS + NI].astype(np.int64), hdr[H_HP:H_HP + NI + 1].astype(np.int64)
        counts = hdr[32:32 + 32 * 16:16].astype(np.int64)
        assert counts.sum() == nv and (occ <= first[1:] - first[:NI]).all() and occ.sum() == nv, "rows held != V (spill overflow?)"
        assert (occ[:32] <= counts).all()
        # the tables against their definition (p3d_mc.hip: tail_tables)
        end = first[:NI] + occ
        assert (ts == np.maximum(first[:NI], nv)).all() and (hs == np.minimum(end, nv)).all()
        tl, hl = np.maximum(end, nv) - ts, np.minimum(first[1:], nv) - hs
        assert (hdr[H_TE:H_TE + NI].astype(np.int64) == ts + tl).all()
        assert (tp == np.cumsum(tl) - tl).all() and (hp[:NI] == np.cumsum(hl) - hl).all() and hp[NI] == hl.sum() == tl.sum()

        def to_dense(row):   # ids of a unit, one by one (a unit's run may be split by the move)
            out = row.copy()
            m = row >= nv
            r = row[m]
            j = np.searchsorted(first[:NI], r, side="right") - 1
            assert (r < first[j] + occ[j]).all(), "an id beyond its interval's rows"
            kk = tp[j] + (r - ts[j])
            i = np.searchsorted(hp[:NI], kk, side="right") - 1
            out[m] = hs[i] + (kk - hp[i])
            return out
    else:
        to_dense = None
    offy = (rec[..., 1] & 0xFFFF).astype(np.int64)
    offz = (rec[..., 1] >> 16).astype(np.int64)
    for axis, (cr, off) in enumerate(((cx, None), (cy, offy), (cz, offz))):
        cu = cr.reshape(rx, ry, ncz, 64)
        rank = np.cumsum(cu, axis=-1) - cu
        vid = base[..., None] + rank + (0 if off is None else off[..., None])
        sel = cu
        v = vid[sel]
        if to_dense is not None:
            v = to_dense(v)
        k = lin.reshape(rx, ry, ncz, 64)[sel] * 3 + axis
        assert v.size == 0 or (v.min() >= 0 and v.max() < nv), (axis, v.min() if v.size else None, nv)
        assert (keys[v] == -1).all(), "vertex id assigned twice"
        keys[v] = k
        # the packed offsets must equal the crossing counts that precede the axis group
        nx = cx.reshape(rx, ry, ncz, 64).sum(-1)
        ny = cy.reshape(rx, ry, ncz, 64).sum(-1)
        has = cu.any(-1)
        if axis == 1:
            assert (offy[has] == nx[has]).all()
        if axis == 2:
            assert (offz[has] == (nx + ny)[has]).all()
    assert (keys >= 0).all(), "some vertex id has no owning edge"
    return keys
