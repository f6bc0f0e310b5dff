"""Generates the committed golden vectors from the oracle (run from the repo root:
`python tests/golden/make_golden.py`).  The reference cannot produce vectors (no tests, TensorFlow not
importable -- SURVEY.md 8c), so these are BUILD-GENERATED: they pin the oracle against silent drift
and give the GPU tests fixed expected values.  Parameters are regenerated from the seed
(oracle.ref_torch.init_sngan_params), only inputs and small outputs are stored."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ref_ops as R  # noqa: E402
from oracle import ref_torch as T  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def bf16_round(a):
    return torch.tensor(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float64).numpy()


def ops():
    rng = np.random.default_rng(2024)
    x = bf16_round(rng.normal(size=(2, 4, 4, 64)))
    w = bf16_round(rng.normal(size=(3, 3, 64, 32)) / 24.)
    b = rng.normal(size=32).astype(np.float32).astype(np.float64)
    dy = bf16_round(rng.normal(size=(2, 4, 4, 32)))
    y = R.conv2d_same(x, w, b)
    dx, dw, db = R.conv2d_same_grads(x, w, dy)
    W = (rng.normal(size=(27, 16)) * 0.1).astype(np.float32).astype(np.float64)
    u = rng.normal(size=(1, 16)).astype(np.float32).astype(np.float64)
    G = rng.normal(size=(27, 16)).astype(np.float32).astype(np.float64)
    Wb, u1, sigma, _ = R.sn_forward(W, u)
    dW = R.sn_backward(W, u, G)
    cx = bf16_round(rng.normal(size=(4, 2, 2, 64)) + 0.5)
    lab = np.array([3, 0, 3, 9])
    gam = (1 + 0.2 * rng.normal(size=(10, 64))).astype(np.float32).astype(np.float64)
    bet = (0.1 * rng.normal(size=(10, 64))).astype(np.float32).astype(np.float64)
    cy, cache = R.cond_batchnorm_forward(cx, lab, gam, bet, groups=2)
    cdy = bf16_round(rng.normal(size=cx.shape))
    cdx, cdg, cdb = R.cond_batchnorm_backward(cdy, lab, gam, cache, groups=2)
    np.savez_compressed(os.path.join(OUT, "ops.npz"), conv_x=x, conv_w=w, conv_b=b, conv_dy=dy, conv_y=y, conv_dx=dx,
                        conv_dw=dw, conv_db=db, sn_W=W, sn_u=u, sn_G=G, sn_Wbar=Wb, sn_u1=u1, sn_sigma=sigma, sn_dW=dW,
                        cbn_x=cx, cbn_labels=lab, cbn_gamma=gam, cbn_beta=bet, cbn_y=cy, cbn_dy=cdy, cbn_dx=cdx,
                        cbn_dgamma=cdg, cbn_dbeta=cdb)


def network():
    seed = 3
    P = T.to_torch(T.init_sngan_params(seed))
    rng = np.random.default_rng(99)
    z = bf16_round(rng.normal(size=(4, 128)))
    labels = np.array([1, 7, 7, 2])
    real_u8 = rng.integers(0, 256, (4, 3072))
    zt, lt = torch.tensor(z), torch.tensor(labels)
    img = T.generator(P, zt, lt, groups=2)
    real = T.preprocess_real(torch.tensor(real_u8), torch.zeros(4, 3072, dtype=torch.float64), torch.float64)
    real = torch.tensor(bf16_round(real.numpy()))
    both = torch.cat([real, torch.tensor(bf16_round(img.detach().numpy()))], 0)
    logits, new_u = T.discriminator(P, both, torch.cat([lt, lt]))
    loss = torch.relu(1. - logits[:4]).mean() + torch.relu(1. + logits[4:]).mean()
    names = ['Discriminator/D.Block.1.Conv1/Filters', 'Discriminator/D.Block.3.Conv2/Filters', 'Discriminator/D.Output/W',
             'Discriminator/D.Embedding_y/W', 'Discriminator/D.Block.2.Shortcut/Biases']
    grads = torch.autograd.grad(loss, [P[k] for k in names])
    np.savez_compressed(os.path.join(OUT, "network.npz"), seed=seed, z=z, labels=labels, real_u8=real_u8,
                        img_head=img.detach().numpy()[:, :96], img_abs_mean=float(img.abs().mean()),
                        logits=logits.detach().numpy(), d_loss=float(loss),
                        u_new_D_Output=new_u['Discriminator/D.Output/spectral_norm/u'].numpy(),
                        grad_names=np.array(names), grad_norms=np.array([float(g.norm()) for g in grads]),
                        grad_D_Output_W=grads[2].numpy())


def configs45():
    """PGGAN (oracle/ref_pggan.py) and Pix2Pix (oracle/ref_pix2pix.py) restatements at small sizes: inputs and small outputs"""
    from oracle import ref_pggan as G
    from oracle import ref_pix2pix as X
    rng = np.random.default_rng(45)
    out = {}
    # PGGAN, 8x8 with the new block fading in (block_count 1, trans)
    P = T.to_torch(G.init_params(5, 1, True, z_dim=32))
    z = bf16_round(rng.normal(size=(3, 32)))
    img = G.generator(P, torch.tensor(z), 0.3, 1, True)
    lg, new_u = G.discriminator(P, img.detach(), 0.3, 1, True, update_u=True)
    dl, _ = G.d_loss(P, img.detach() * 0.5, torch.tensor(z), 0.3, 1, True)
    out.update(pg_z=z, pg_img=img.detach().numpy(), pg_logits=lg.detach().numpy(), pg_d_loss=float(dl),
               pg_u=new_u['d_net/D.Conv/filters/spectral_norm/u'].numpy(), pg_mbstd=G.minibatch_std_numpy(z.reshape(3, 2, 2, 8)),
               pg_resize=G.resize_bilinear(z.reshape(1, 4, 8, 3), (8, 16)))
    # Pix2Pix: the general convolution and instance norm, and the PatchGAN critic on 64x64 (ndf 8)
    x = bf16_round(rng.normal(size=(2, 6, 6, 5)))
    w = bf16_round(rng.normal(size=(4, 4, 5, 3)) / 9.)
    out.update(px_x=x, px_w=w,
               px_s2_same=X.conv2d_tf(torch.tensor(x), torch.tensor(w), None, 2, 'SAME').numpy(),
               px_s1_same=X.conv2d_tf(torch.tensor(x), torch.tensor(w), None, 1, 'SAME').numpy(),
               px_s1_valid=X.conv2d_tf(torch.tensor(x), torch.tensor(w), None, 1, 'VALID', 1).numpy(),
               px_inorm=X.instance_norm(torch.tensor(x), torch.ones(1, 5, dtype=torch.float64) * 1.5, torch.ones(1, 5, dtype=torch.float64) * 0.1).numpy())
    Pd = T.to_torch(X.init_params(9, ngf=8, ndf=8))
    a = bf16_round(rng.uniform(-1, 1, size=(2, 64, 64, 3)))
    b = bf16_round(rng.uniform(-1, 1, size=(2, 64, 64, 3)))
    pr, _ = X.discriminator(Pd, torch.tensor(a), torch.tensor(b))
    out.update(px_a=a, px_b=b, px_patch=pr.detach().numpy())
    np.savez_compressed(os.path.join(OUT, "configs45.npz"), **out)


if __name__ == "__main__":
    ops()
    network()
    configs45()
    print("golden vectors written to", OUT)
