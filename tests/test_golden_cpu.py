"""CPU: the oracle reproduces the committed golden vectors (guards against silent oracle drift)."""
import os

import numpy as np
import torch

from oracle import ref_ops as R
from oracle import ref_torch as T


def test_ops_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "ops.npz"))
    np.testing.assert_allclose(R.conv2d_same(g["conv_x"], g["conv_w"], g["conv_b"]), g["conv_y"], atol=1e-12)
    dx, dw, db = R.conv2d_same_grads(g["conv_x"], g["conv_w"], g["conv_dy"])
    np.testing.assert_allclose(dx, g["conv_dx"], atol=1e-12)
    np.testing.assert_allclose(dw, g["conv_dw"], atol=1e-11)
    Wb, u1, sigma, _ = R.sn_forward(g["sn_W"], g["sn_u"])
    np.testing.assert_allclose(Wb, g["sn_Wbar"], atol=1e-13)
    np.testing.assert_allclose(R.sn_backward(g["sn_W"], g["sn_u"], g["sn_G"]), g["sn_dW"], atol=1e-12)
    y, cache = R.cond_batchnorm_forward(g["cbn_x"], g["cbn_labels"], g["cbn_gamma"], g["cbn_beta"], groups=2)
    np.testing.assert_allclose(y, g["cbn_y"], atol=1e-12)
    dxc, dg, dbt = R.cond_batchnorm_backward(g["cbn_dy"], g["cbn_labels"], g["cbn_gamma"], cache, groups=2)
    np.testing.assert_allclose(dxc, g["cbn_dx"], atol=1e-11)
    np.testing.assert_allclose(dg, g["cbn_dgamma"], atol=1e-11)


def test_network_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "network.npz"))
    P = T.to_torch(T.init_sngan_params(int(g["seed"])))
    img = T.generator(P, torch.tensor(g["z"]), torch.tensor(g["labels"]), groups=2).detach().numpy()
    np.testing.assert_allclose(img[:, :96], g["img_head"], atol=1e-10)
    np.testing.assert_allclose(np.abs(img).mean(), g["img_abs_mean"], atol=1e-10)


def test_pggan_pix2pix_golden(golden_dir):
    """oracle/ref_pggan.py and oracle/ref_pix2pix.py reproduce tests/golden/configs45.npz"""
    from oracle import ref_pggan as G
    from oracle import ref_pix2pix as X
    g = np.load(os.path.join(golden_dir, "configs45.npz"))
    P = T.to_torch(G.init_params(5, 1, True, z_dim=32))
    z = torch.tensor(g["pg_z"])
    img = G.generator(P, z, 0.3, 1, True)
    np.testing.assert_allclose(img.detach().numpy(), g["pg_img"], atol=1e-10)
    lg, new_u = G.discriminator(P, img.detach(), 0.3, 1, True, update_u=True)
    np.testing.assert_allclose(lg.detach().numpy(), g["pg_logits"], atol=1e-10)
    np.testing.assert_allclose(new_u['d_net/D.Conv/filters/spectral_norm/u'].numpy(), g["pg_u"], atol=1e-12)
    dl, _ = G.d_loss(P, img.detach() * 0.5, z, 0.3, 1, True)
    assert abs(float(dl) - float(g["pg_d_loss"])) < 1e-10
    np.testing.assert_allclose(G.minibatch_std_numpy(g["pg_z"].reshape(3, 2, 2, 8)), g["pg_mbstd"], atol=1e-12)
    np.testing.assert_allclose(G.resize_bilinear(g["pg_z"].reshape(1, 4, 8, 3), (8, 16)), g["pg_resize"], atol=1e-12)
    x, w = torch.tensor(g["px_x"]), torch.tensor(g["px_w"])
    np.testing.assert_allclose(X.conv2d_tf(x, w, None, 2, 'SAME').numpy(), g["px_s2_same"], atol=1e-12)
    np.testing.assert_allclose(X.conv2d_tf(x, w, None, 1, 'SAME').numpy(), g["px_s1_same"], atol=1e-12)
    np.testing.assert_allclose(X.conv2d_tf(x, w, None, 1, 'VALID', 1).numpy(), g["px_s1_valid"], atol=1e-12)
    np.testing.assert_allclose(X.conv2d_numpy(g["px_x"], g["px_w"], None, 2, 1, (3, 3)), g["px_s2_same"], atol=1e-12)
    np.testing.assert_allclose(X.instance_norm(x, torch.ones(1, 5, dtype=torch.float64) * 1.5, torch.ones(1, 5, dtype=torch.float64) * 0.1).numpy(),
                               g["px_inorm"], atol=1e-12)
    Pd = T.to_torch(X.init_params(9, ngf=8, ndf=8))
    pr, _ = X.discriminator(Pd, torch.tensor(g["px_a"]), torch.tensor(g["px_b"]))
    np.testing.assert_allclose(pr.detach().numpy(), g["px_patch"], atol=1e-10)
