"""Thin torch-tensor wrappers around the C ABI for the GPU tests (all calls go through ctypes)."""
import ctypes as C

import torch

from refil_amd import _lib
from refil_amd._lib import AttnDesc, AttnQkvDesc, GemmDesc, GruDesc, RowMap, check, lib, ptr


def _stream():
    return _lib.current_stream_ptr()


def gemm(A, B, C_out, M, N, K, lda, ldb, ldc, flags=0, bias=None, aux=None, rowmask=None, rowmask_mod=0,
         colsum=None, partial=None, batch=1, splits=1, sA=0, sB=0, sC=0, sBias=0, sColsum=0,
         a_map=(0, 0, 0), b_map=(0, 0, 0), c_map=(0, 0, 0), row_index=None, row_count=None):
    d = GemmDesc()
    d.A, d.B, d.C = A.data_ptr(), B.data_ptr(), C_out.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    d.aux = aux.data_ptr() if aux is not None else None
    d.rowmask = rowmask.data_ptr() if rowmask is not None else None
    d.colsum = colsum.data_ptr() if colsum is not None else None
    d.partial = partial.data_ptr() if partial is not None else None
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc = M, N, K, lda, ldb, ldc
    d.sA, d.sB, d.sC, d.sBias, d.sColsum = sA, sB, sC, sBias, sColsum
    d.a_map, d.b_map, d.c_map = RowMap(*a_map), RowMap(*b_map), RowMap(*c_map)
    d.rowmask_mod, d.batch, d.splits, d.flags = rowmask_mod, batch, splits, flags
    if row_index is not None:
        d.row_index, d.row_count = row_index.data_ptr(), row_count.data_ptr()
    check(lib().refil_gemm(C.byref(d), _stream()), "refil_gemm")


def attn_desc(Q, K, V, ldq, ldkv, R, T1, ne, na, heads, hd, variants, obs_mask=None, ent_mask=None, ent_mask0=None,
              group_bits=None):
    d = AttnDesc()
    d.Q, d.K, d.V, d.ldq, d.ldkv = Q.data_ptr(), K.data_ptr(), V.data_ptr(), ldq, ldkv
    d.R, d.T1, d.ne, d.na, d.heads, d.hd = R, T1, ne, na, heads, hd
    d.nvar = len(variants)
    for i, v in enumerate(variants):
        d.var[i] = v
    if obs_mask is not None:
        d.obs_mask, d.om_sB, d.om_sT = obs_mask.data_ptr(), obs_mask.stride(0), obs_mask.stride(1)
    d.ent_mask = ent_mask.data_ptr() if ent_mask is not None else None
    d.ent_mask0 = ent_mask0.data_ptr() if ent_mask0 is not None else None
    d.group_bits = group_bits.data_ptr() if group_bits is not None else None
    d._keep = [Q, K, V, obs_mask, ent_mask, ent_mask0, group_bits]   # the desc only holds raw addresses
    return d


def attn_skip(d, t_last=None, kv_dead=None, q_dead=None):
    d._keep += [t_last, kv_dead, q_dead]
    d.t_last = t_last.data_ptr() if t_last is not None else None
    d.kv_dead = kv_dead.data_ptr() if kv_dead is not None else None
    d.q_dead = q_dead.data_ptr() if q_dead is not None else None
    return d


def attn_forward(d: AttnDesc, O, ldo, sO):
    d.O, d.ldo, d.sO = O.data_ptr(), ldo, sO
    check(lib().refil_attn_forward(C.byref(d), _stream()), "refil_attn_forward")


def attn_qkv_forward(d: AttnDesc, X, ldx, W_in, O, ldo, sO, q_out=None, k_out=None, v_out=None):
    """in_trans + attention core in one launch (refil_attn_qkv_forward); d carries rows / masks / precomputed mask words."""
    d.O, d.ldo, d.sO = O.data_ptr(), ldo, sO
    qd = AttnQkvDesc()
    qd.attn = d
    qd.X, qd.ldx, qd.W_in = X.data_ptr(), ldx, W_in.data_ptr()
    qd.q_out = q_out.data_ptr() if q_out is not None else None
    qd.k_out = k_out.data_ptr() if k_out is not None else None
    qd.v_out = v_out.data_ptr() if v_out is not None else None
    check(lib().refil_attn_qkv_forward(C.byref(qd), _stream()), "refil_attn_qkv_forward")


def attn_backward(d: AttnDesc, dO, ldo, sO, dQ, dK, dV):
    d._keep.append(dO)
    d.dO, d.ldo, d.sO = dO.data_ptr(), ldo, sO
    d.dQ, d.dK, d.dV = dQ.data_ptr(), dK.data_ptr(), dV.data_ptr()
    check(lib().refil_attn_backward(C.byref(d), _stream()), "refil_attn_backward")


def pool_forward(d: AttnDesc, mode, O, ldo, sO):
    d.O, d.ldo, d.sO = O.data_ptr(), ldo, sO
    check(lib().refil_pool_forward(C.byref(d), mode, _stream()), "refil_pool_forward")


def pool_backward(d: AttnDesc, mode, dO, ldo, sO, dK):
    d._keep.append(dO)
    d.dO, d.ldo, d.sO = dO.data_ptr(), ldo, sO
    d.dK = dK.data_ptr()
    check(lib().refil_pool_backward(C.byref(d), mode, _stream()), "refil_pool_backward")


def gru_desc(gi, hsx, w_hh, b_hh, NR, T1, na, H=64, saves=None, dhs=None, dgi=None, dgh=None):
    d = GruDesc()
    d.gi = gi.data_ptr() if gi is not None else None
    d.hsx, d.w_hh, d.b_hh = hsx.data_ptr(), w_hh.data_ptr(), b_hh.data_ptr()
    if saves is not None:
        d.save_r, d.save_z, d.save_n, d.save_ghn = [s.data_ptr() for s in saves]
    if dhs is not None:
        d.dhs, d.dgi, d.dgh = dhs.data_ptr(), dgi.data_ptr(), dgh.data_ptr()
    d.NR, d.T1, d.na, d.H = NR, T1, na, H
    d._keep = [gi, hsx, w_hh, b_hh, saves, dhs, dgi, dgh]
    return d


def gru_skip(d, t_last, B):
    d._keep.append(t_last)
    d.t_last, d.B = t_last.data_ptr(), B
    return d


def gru_forward(d):
    check(lib().refil_gru_forward(C.byref(d), _stream()), "refil_gru_forward")


def gru_backward(d):
    check(lib().refil_gru_backward(C.byref(d), _stream()), "refil_gru_backward")


def attn_mask_words(d: AttnDesc, na):
    """precompute the mask words of desc's variants (refil_attn_mask_words) and attach them to the desc"""
    na_pad = (na + 15) // 16 * 16
    dev = d._keep[0].device       # (the tensors the desc was built from: the GPU in the gpu tier, host memory under tests/emu)
    mw = torch.zeros(d.R * d.nvar * na_pad, dtype=torch.int64, device=dev)
    rb = torch.zeros(d.R * 3, dtype=torch.int64, device=dev)
    check(lib().refil_attn_mask_words(C.byref(d), ptr(mw), ptr(rb), _stream()), "refil_attn_mask_words")
    d._keep += [mw, rb]
    d.mask_words, d.row_bits, d.mask_words_nvar = mw.data_ptr(), rb.data_ptr(), d.nvar
    return d
