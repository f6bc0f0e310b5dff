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
