#!/usr/bin/env python
"""Groundwork for DESIGN.md "What comes next" item 4 (NOT used by the product): the segmentation branch's algebra one decoder level
further, checked in fp64 on the CPU against plain torch.

Today (NNDET_SEG_UP) the branch's top-down term is  zup[m][pi] = conv3_half(x1; Wc)[m][pi]  with x1 = lat1(a1) + up2(x2) formed by the
decoder (lateral P1 = 1x1x1 conv of the encoder's half-resolution output a1, up2 = k = s = 2 transposed conv of the quarter-resolution
map x2) and Wc = up_compose(wc, W_up1, ...) [8 parity classes, I channels, 27 taps]. x1 has no other reader, so

    conv3_half(x1; Wc)[m][pi] = conv3_half(a1; WcA)[m][pi]                        WcA[pi][j][d] = sum_i Wc[pi][i][d] W_lat1[i][j]
                              + conv3_quarter(x2; Wcc)[r][(rho, pi)]              m = 2 r + rho;  Wcc[., pi] = up_compose(Wc[pi] as a tap-major
                              + cb1[pi][border class of m at half resolution]                      kernel over the 27 offsets d, W_up2, b1)
    b1 = b_lat1 + b_up2

i.e. the SAME composition (arch/segmenter.py: up_compose / up_param_grads) applied once per parity class pi with "wc := Wc[pi]": a
quarter-resolution 3x3x3 convolution J -> 64 (8 x 8 classes) plus a half-resolution one on a1, and the backward pass follows from their
weight gradients by the same two helpers. This script builds both sides for random fp64 tensors and prints the largest deviations of
the forward value and of every gradient (x2, a1, W_lat1, b_lat1, W_up2, b_up2, Wc)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from nndetection_amd.arch.segmenter import up_compose, up_param_grads


def border_classes(n):
    c = torch.ones(n, dtype=torch.long)
    c[0], c[-1] = 0, 2
    return c


def s2d(t):            # [N, C, 2D, 2H, 2W] -> [N, 8 * C (parity-major), D, H, W]
    N, C, D2, H2, W2 = t.shape
    return t.reshape(N, C, D2 // 2, 2, H2 // 2, 2, W2 // 2, 2).permute(0, 3, 5, 7, 1, 2, 4, 6).reshape(N, 8 * C, D2 // 2, H2 // 2, W2 // 2)


def d2s(t, C):         # inverse of s2d
    N, _, D, H, W = t.shape
    return t.reshape(N, 2, 2, 2, C, D, H, W).permute(0, 4, 5, 1, 6, 2, 7, 3).reshape(N, C, 2 * D, 2 * H, 2 * W)


def main():
    torch.manual_seed(0)
    dd = torch.float64
    N, J, I, A1, D4, H4, W4 = 2, 5, 6, 4, 2, 3, 2                 # x2: J channels at quarter res; x1: I channels; a1: A1 channels
    D2, H2, W2 = 2 * D4, 2 * H4, 2 * W4
    x2 = torch.randn(N, J, D4, H4, W4, dtype=dd, requires_grad=True)
    a1 = torch.randn(N, A1, D2, H2, W2, dtype=dd, requires_grad=True)
    w_lat1 = torch.randn(I, A1, 1, 1, 1, dtype=dd, requires_grad=True)
    b_lat1 = torch.randn(I, dtype=dd, requires_grad=True)
    w_up2 = torch.randn(J, I, 2, 2, 2, dtype=dd, requires_grad=True)
    b_up2 = torch.randn(I, dtype=dd, requires_grad=True)
    Wc = torch.randn(8, I, 3, 3, 3, dtype=dd, requires_grad=True)     # what up_compose(wc, W_up1, .) hands to the half-resolution convolution
    G = torch.randn(N, 8, D2, H2, W2, dtype=dd)                       # dL/dzup
    # ---- reference: the decoder forms x1
    x1 = F.conv3d(a1, w_lat1, b_lat1) + F.conv_transpose3d(x2, w_up2, b_up2, stride=2)
    zr = F.conv3d(x1, Wc, padding=1)
    (zr * G).sum().backward()
    # ---- composed: x1 never exists
    with torch.no_grad():
        WcA = torch.einsum("pixyz,ij->pjxyz", Wc, w_lat1.reshape(I, A1))
        zA = F.conv3d(a1, WcA, padding=1)
        b1 = b_lat1 + b_up2
        Wcc, cb1 = [], []
        for pi in range(8):
            wck = Wc[pi].reshape(I, 27).t().contiguous()              # tap-major [27 offsets, I]: the "wc" of the inner composition
            w, cb = up_compose(wck, w_up2.detach(), b1)
            Wcc.append(w)                                             # [8 (rho), J, 3, 3, 3]
            cb1.append(cb.reshape(3, 3, 3))
        Wcc64 = torch.stack(Wcc, dim=1).reshape(64, J, 3, 3, 3)       # output channel (rho, pi)
        zB4 = F.conv3d(x2, Wcc64, padding=1)                          # [N, 64, D4, H4, W4]
        zB = d2s(zB4, 8)                                              # channel rho * 8 + pi -> [N, 8 (pi), D2, H2, W2]
        cd, ch, cw = border_classes(D2), border_classes(H2), border_classes(W2)
        cbt = torch.stack([c[cd][:, ch][:, :, cw] for c in cb1], 0)   # [8, D2, H2, W2]
        z = zA + zB + cbt.unsqueeze(0)
        print(f"forward: max |composed - reference| = {float((z - zr).abs().max()):.3e}")
        # ---- backward from G
        da1 = torch.nn.grad.conv3d_input(a1.shape, WcA, G, padding=1)
        dWcA = torch.nn.grad.conv3d_weight(a1.detach(), WcA.shape, G, padding=1)
        dw_lat1 = torch.einsum("pixyz,pjxyz->ij", Wc, dWcA).reshape(I, A1, 1, 1, 1)
        dWc = torch.einsum("ij,pjxyz->pixyz", w_lat1.reshape(I, A1), dWcA)
        G4 = s2d(G)                                                   # channel rho * 8 + pi, like Wcc64's outputs
        dx2 = torch.nn.grad.conv3d_input(x2.shape, Wcc64, G4, padding=1)
        dWcc = torch.nn.grad.conv3d_weight(x2.detach(), Wcc64.shape, G4, padding=1).reshape(8, 8, J, 3, 3, 3)   # [rho, pi, J, ...]
        dw_up2 = torch.zeros_like(w_up2)
        db1 = torch.zeros(I, dtype=dd)
        cls_idx = (cd[:, None, None] * 3 + ch[None, :, None]) * 3 + cw[None, None, :]
        for pi in range(8):
            wck = Wc[pi].reshape(I, 27).t().contiguous()
            csum = torch.zeros(27, dtype=dd).index_add_(0, cls_idx.reshape(-1), G[:, pi].sum(0).reshape(-1))      # half-res border-class sums of dzup[:, pi]
            dwu, dbs, ec = up_param_grads(wck, w_up2.detach(), b1, dWcc[:, pi], csum)
            dw_up2 += dwu
            db1 += dbs
            dWc[pi] += ec.t().reshape(I, 3, 3, 3)
        for name, ours, ref in (("dx2", dx2, x2.grad), ("da1", da1, a1.grad), ("dW_lat1", dw_lat1, w_lat1.grad), ("db_lat1", db1, b_lat1.grad),
                                ("dW_up2", dw_up2, w_up2.grad), ("db_up2", db1, b_up2.grad), ("dWc", dWc, Wc.grad)):
            print(f"{name:8s}: max |composed - reference| = {float((ours - ref).abs().max()):.3e}   (max |reference| {float(ref.abs().max()):.3e})")


if __name__ == "__main__":
    main()
