"""Pin the CPU oracle against outputs of the real reference (tests/golden/*.npz,
made by oracle/make_golden.py from /root/reference/models/wesup.py)."""
import glob
import os

import numpy as np
import pytest
import torch

from oracle import wesup_oracle as orc

CASES = ['c32_point', 'c32_point_far', 'c64_point_tie', 'c64_full', 'c96x80_point', 'c64_identical']


def load_case(golden_dir, name):
    fx = dict(np.load(os.path.join(golden_dir, name + '.npz')))
    shape = tuple(int(v) for v in fx['mask_shape'])
    fx['mask'] = np.unpackbits(fx['mask'])[:int(np.prod(shape))].reshape(shape)
    return fx


def rel_err(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / (np.abs(b).max() + 1e-30))


@pytest.mark.parametrize('name', CASES)
def test_oracle_matches_reference(golden_dir, name):
    fx = load_case(golden_dir, name)
    torch.set_num_threads(8)
    weights = orc.make_weights(int(fx['seed']), feat_scale=float(fx['feat_scale']))
    seg = torch.from_numpy(fx['seg'].astype(np.int64))
    mask = torch.from_numpy(fx['mask'].astype(np.int64))
    img = torch.from_numpy(fx['img'])

    # --- preprocess: integer outputs bit-exact (dense restatement and label-map restatement)
    sp_maps, sp_labels = orc.preprocess_superpixels_dense(seg, mask)
    assert np.array_equal(sp_labels.numpy(), fx['sp_labels'])
    assert np.array_equal(sp_maps.argmax(dim=0).numpy(), fx['new_row'])
    pp = orc.preprocess_superpixels(seg, mask)
    assert np.array_equal(pp['sp_labels'].numpy(), fx['sp_labels'])
    assert np.array_equal(pp['inv_perm'][seg].numpy(), fx['new_row'])
    np.testing.assert_allclose(1.0 / pp['area'][pp['perm']].float().numpy(), fx['sp_maps_max'], rtol=1e-6)

    # --- forward + loss + backward through the label-map restatement
    loss, grads, _, _, outs, mets = orc.train_step(weights, fx['img'][None], fx['seg'][None].astype(np.int64),
                                                   fx['mask'][None].astype(np.int64))
    o = outs[0]
    fm = o['fm'].detach()
    assert rel_err(fm.mean(dim=(1, 2)).numpy(), fx['fm_chan_mean']) < 1e-5
    assert rel_err(fm[::37, ::5, ::7].numpy(), fx['fm_sample']) < 1e-5
    assert rel_err(o['sp_features'].detach().numpy(), fx['sp_features']) < 1e-4
    assert rel_err(o['sp_pred'].detach().numpy(), fx['sp_pred']) < 1e-5
    assert rel_err(o['pred'].detach().numpy()[None], fx['pred']) < 1e-5
    assert np.array_equal(o['pred'].detach().round().long().numpy()[None], fx['post_pred'])
    assert abs(loss - float(fx['loss'])) <= 1e-4 * abs(float(fx['loss']))

    # --- propagation: argmax indices and pseudo labels bit-exact
    n_l = pp['n_l']
    if n_l < pp['K']:
        y_u, W_ul, max_sim, src = orc.label_propagate(torch.from_numpy(fx['sp_features']),
                                                      torch.from_numpy(fx['sp_labels']), 0.8, return_aux=True)
        assert np.array_equal(src.numpy(), fx['src'])
        assert np.array_equal(y_u.numpy(), fx['y_u'])
        assert rel_err(W_ul.numpy(), fx['W_ul']) < 1e-5
        assert mets[0]['propagated_labels'] == float(fx['propagated_labels'])
        assert abs(mets[0]['propagate_loss'] - float(fx['propagate_loss'])) < 1e-5
        assert abs(mets[0]['labeled_sp_ratio'] - float(fx['labeled_sp_ratio'])) < 1e-12

    # --- gradients
    for k in [k[6:] for k in fx if k.startswith('gnorm.')]:
        g = grads[k]
        ref_norm = float(fx['gnorm.' + k])
        assert abs(g.double().norm().item() - ref_norm) <= 2e-4 * ref_norm + 1e-12, k
        samp = g.flatten()[::max(1, g.numel() // 64)][:64].numpy()
        assert np.abs(samp - fx['gsamp.' + k]).max() <= 2e-4 * (np.abs(fx['gsamp.' + k]).max() + 1e-12), k


@pytest.mark.parametrize('name', ['c480_g14', 'c800_g39'])
def test_oracle_matches_reference_at_full_size(golden_dir, name):
    """BASELINE configs[0] at its real size (480x480, 196 superpixels) and one image of configs[3] (800x800, 1521
    superpixels): the oracle against the reference's own outputs (models/wesup.py:18-63, 263-304, 492-531 + backward).
    c480_g24 (576 superpixels, one image of configs[1]) is checked on the GPU box together with the HIP path
    (tests/test_fullsize_gpu.py)."""
    from conftest import load_full_size_case
    fx = load_full_size_case(golden_dir, name)
    torch.set_num_threads(8)
    weights = orc.make_weights(int(fx['seed']), feat_scale=float(fx['feat_scale']))
    seg = torch.from_numpy(fx['seg'].astype(np.int64))
    pp = orc.preprocess_superpixels(seg, torch.from_numpy(fx['mask'].astype(np.int64)))
    assert np.array_equal(pp['sp_labels'].numpy(), fx['sp_labels'])
    assert np.array_equal(pp['inv_perm'][seg].numpy().astype(np.int16), fx['new_row'])
    loss, grads, _, _, outs, mets = orc.train_step(weights, fx['img'][None], fx['seg'][None].astype(np.int64),
                                                   fx['mask'][None].astype(np.int64))
    o = outs[0]
    fm = o['fm'].detach()
    assert rel_err(fm.mean(dim=(1, 2)).numpy(), fx['fm_chan_mean']) < 1e-5
    assert rel_err(fm[::97, ::23, ::29].numpy(), fx['fm_sample']) < 1e-5
    assert rel_err(o['sp_features'].detach().numpy(), fx['sp_features']) < 1e-4
    assert rel_err(o['sp_pred'].detach().numpy(), fx['sp_pred']) < 1e-4
    assert rel_err(o['pred'].detach().numpy()[::7, ::11], fx['pred_sample']) < 1e-4
    assert np.array_equal(o['pred'].detach().round().long().numpy()[None], fx['post_pred'])
    assert abs(loss - float(fx['loss'])) <= 1e-4 * abs(float(fx['loss']))
    y_u, W_ul, max_sim, src = orc.label_propagate(torch.from_numpy(fx['sp_features']), torch.from_numpy(fx['sp_labels']),
                                                  0.8, return_aux=True)
    assert np.array_equal(src.numpy(), fx['src']) and np.array_equal(y_u.numpy(), fx['y_u'])
    assert rel_err(max_sim.numpy(), fx['max_sim']) < 1e-5 and rel_err(W_ul.sum(dim=1).numpy(), fx['W_ul_rowsum']) < 1e-5
    assert mets[0]['propagated_labels'] == float(fx['propagated_labels'])
    assert abs(mets[0]['propagate_loss'] - float(fx['propagate_loss'])) < 1e-5
    for k in [k[6:] for k in fx if k.startswith('gnorm.')]:          # every parameter of the model
        ref_norm = float(fx['gnorm.' + k])
        assert abs(grads[k].double().norm().item() - ref_norm) <= 2e-4 * ref_norm + 1e-12, k
        samp = grads[k].flatten()[::max(1, grads[k].numel() // 64)][:64].numpy()
        assert np.abs(samp - fx['gsamp.' + k]).max() <= 3e-4 * float(fx['gmax.' + k]), k


def test_faithful_variant_matches_reference(golden_dir):
    """The "faithful" CPU-baseline variant (dense maps, incremental cat, dense mm, argmax paint-back) against the
    reference's outputs, so that what bench.py times as the reference's CPU path IS the reference's algorithm."""
    fx = load_case(golden_dir, 'c96x80_point')
    w = orc.to_torch(orc.make_weights(int(fx['seed']), feat_scale=float(fx['feat_scale'])))
    sp_maps, sp_labels = orc.preprocess_superpixels_dense(torch.from_numpy(fx['seg'].astype(np.int64)),
                                                          torch.from_numpy(fx['mask'].astype(np.int64)))
    o = orc.forward_image_faithful(w, torch.from_numpy(fx['img']), sp_maps)
    assert rel_err(o['fm'][::37, ::5, ::7].numpy(), fx['fm_sample']) < 1e-5
    assert rel_err(o['sp_features'].numpy(), fx['sp_features']) < 1e-4
    assert rel_err(o['pred'].numpy()[None], fx['pred']) < 1e-5
    loss = orc.compute_loss(o['sp_pred'], o['sp_features'], sp_labels)
    assert abs(float(loss) - float(fx['loss'])) <= 1e-4 * abs(float(fx['loss']))


def test_reference_behaviours(golden_dir):
    fx = np.load(os.path.join(golden_dir, 'behaviours.npz'))
    yu = orc.label_propagate(torch.zeros(5, 32), torch.tensor([[1., 0.], [0., 1.]]), 0.8)
    assert np.array_equal(yu.numpy(), fx['identical_yu'])          # ties -> first labelled index
    ce0 = orc.cross_entropy(torch.tensor([[0.3, 0.7]]), torch.zeros(1, 2))
    assert float(ce0) == float(fx['ce_zero']) == 0.0


def test_empty_id_rejected():
    seg = torch.tensor([[1, 1], [2, 2]])      # 1-based ids: id 0 empty -> NaN row in the reference
    with pytest.raises(ValueError):
        orc.preprocess_superpixels(seg, None)


def test_pixel_inference_oracle_matches_reference(golden_dir):
    """oracle.pixel_inference vs the reference's WESUPPixelInference.forward (models/wesup.py:382-400)."""
    fx = np.load(os.path.join(golden_dir, 'pixel_infer.npz'))
    w = orc.to_torch(orc.make_weights(int(fx['seed']), feat_scale=float(fx['feat_scale'])))
    out = orc.pixel_inference(w, torch.from_numpy(fx['img'])[None])
    assert tuple(out.shape) == fx['out'].shape
    assert rel_err(out.numpy(), fx['out']) < 1e-5


def test_challenge_metrics_match_the_reference(golden_dir):
    """detection_f1 / object_dice / object_hausdorff / hausdorff (utils/metrics.py:48-281) against values produced by
    the reference's own functions (oracle/make_golden.py metrics_golden)."""
    from wesup_amd.utils import metrics as M
    fx = np.load(os.path.join(golden_dir, 'metrics.npz'))
    for i in range(int(fx['n'])):
        S, G, want = fx[f'S{i}'], fx[f'G{i}'], fx[f'v{i}']
        got = [M.detection_f1(S, G), M.object_dice(S, G), M.object_hausdorff(S, G) if S.any() and G.any() else np.nan,
               M.hausdorff(S, G)]
        for g, w in zip(got, want):
            if np.isnan(w):
                continue
            assert (np.isinf(w) and np.isinf(g)) or abs(g - w) <= 1e-9 * max(1.0, abs(w)), (i, got, want)
        # torch inputs are accepted like numpy ones
        assert M.object_dice(torch.from_numpy(S), torch.from_numpy(G)) == got[1]
