"""numpy restatement of the reference's NetVLAD-FC pooling head -- TEST INFRASTRUCTURE ONLY
(see oracle/gloc_oracle.h for the rules).  Follows model/netvlad_fc.py:73-109 (NetVLAD.forward):

  :76-77   x <- x / max(||x||_2 over channels, 1e-12)          (F.normalize, dim=1)
  :80-81   soft_assign = softmax over clusters of the 1x1 conv  (conv weight [K,C], optional bias)
  :88-96   vlad[k,c] = sum_p soft[k,p] * (x[c,p] - centroid[k,c])
  :99      intra-normalisation: vlad[k,:] / max(||vlad[k,:]||, 1e-12)
  :101-102 flatten [K*C], L2-normalise
  :105     @ hidden1_weights [K*C, out]
  :106-107 optional GatingContext (:120-146; off in main.py:594): y * sigmoid((y W) * scale + shift) with
           BatchNorm1d in eval mode folded into scale / shift (gating_forward)

PINNED: checked against the reference module itself (imported in the build container by
tests/golden/make_vlad_goldens.py) through the committed fixtures tests/golden/vlad_*.npz.
All arithmetic in float32, as the reference.
"""
import numpy as np


def netvlad_fc_forward(x, conv_w, conv_b, centroids, fc_w, normalize_input=True):
    """x: [N, C, H, W] (or [N, C, P]) float32.  Returns [N, out] float32."""
    x = np.asarray(x, np.float32)
    N, C = x.shape[:2]
    x = x.reshape(N, C, -1)
    K = conv_w.shape[0]
    if normalize_input:
        nrm = np.sqrt((x * x).sum(axis=1, keepdims=True, dtype=np.float32))
        x = x / np.maximum(nrm, np.float32(1e-12))
    logits = np.einsum("kc,ncp->nkp", conv_w.astype(np.float32), x).astype(np.float32)
    if conv_b is not None:
        logits = logits + conv_b.astype(np.float32)[None, :, None]
    logits = logits - logits.max(axis=1, keepdims=True)
    e = np.exp(logits, dtype=np.float32)
    soft = e / e.sum(axis=1, keepdims=True, dtype=np.float32)
    vlad = np.einsum("nkp,ncp->nkc", soft, x).astype(np.float32) - \
        centroids.astype(np.float32)[None] * soft.sum(axis=2, dtype=np.float32)[:, :, None]
    n1 = np.sqrt((vlad * vlad).sum(axis=2, keepdims=True, dtype=np.float32))
    vlad = vlad / np.maximum(n1, np.float32(1e-12))
    vlad = vlad.reshape(N, K * C)
    n2 = np.sqrt((vlad * vlad).sum(axis=1, keepdims=True, dtype=np.float32))
    vlad = vlad / np.maximum(n2, np.float32(1e-12))
    return (vlad @ fc_w.astype(np.float32)).astype(np.float32)


def gating_forward(y, gating_w, scale, shift):
    """GatingContext.forward (model/netvlad_fc.py:136-146) in eval mode; see gloc_vlad_set_gating."""
    y = np.asarray(y, np.float32)
    g = (y @ gating_w.astype(np.float32)).astype(np.float32) * scale.astype(np.float32) + shift.astype(np.float32)
    return (y * (np.float32(1) / (np.float32(1) + np.exp(-g, dtype=np.float32)))).astype(np.float32)


def fold_batch_norm(weight, bias, running_mean, running_var, eps=1e-5):
    scale = (weight / np.sqrt(running_var + eps)).astype(np.float32)
    return scale, (bias - running_mean * scale).astype(np.float32)
