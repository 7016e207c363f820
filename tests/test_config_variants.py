"""The model variants the reference's shipped configs select besides configs/Ours.yaml:
configs/Ablation_NoCondition_Convolution.yaml (source_condition: false + condition_ablation: the down-sampled q-map is
the FiLM's beta | gamma, model/blocks.py:244-247, transforms.py:31-42) and the ``dense: false`` switch of the synthesis
transform (transforms.py:159-163, blocks.py:168-175).  Same stage-by-stage rule as every codec comparison (tests/_parity.py)."""
import copy

import numpy as np
import pytest
import torch

DEV = "cuda:0"


def _configs(pcc):
    syn = pcc.synthetic
    not_dense = copy.deepcopy(syn.OURS_CONFIG)
    not_dense["g_s"]["dense"] = False
    both = copy.deepcopy(syn.ABLATION_NOCONDITION_CONFIG)
    both["g_s"]["dense"] = False
    return {"no_condition": syn.ABLATION_NOCONDITION_CONFIG, "not_dense": not_dense, "no_condition_not_dense": both}


def test_variant_models_construct_with_the_reference_parameter_sets(pcc):
    syn = pcc.synthetic
    base = {k for k, _ in pcc.ColorModel(syn.OURS_CONFIG).named_parameters()}
    abl = {k for k, _ in pcc.ColorModel(syn.ABLATION_NOCONDITION_CONFIG).named_parameters()}
    # source_condition: false drops exactly the two cond_conv chains (transforms.py:33-41, 165-173); every other
    # parameter — including the predict_layers the ablation never runs — stays, as in the reference
    gone = base - abl
    assert gone and all(".cond_conv." in k for k in gone) and not (abl - base)
    assert {k.split(".cond_conv.")[0] for k in gone} == {"g_a", "g_s"}
    # model/model.py:22-24: an "entropy_model_map" section selects two MeanScaleHyperprior models (tests/test_two_hyperprior.py);
    # against the shipped model: h_q goes, a second h_a / h_s / entropy_bottleneck comes
    two = {k for k, _ in pcc.ColorModel(syn.TWO_HYPERPRIOR_CONFIG).named_parameters()}
    assert all(k.startswith("entropy_model.h_q.") for k in base - two) and base - two
    assert all(k.startswith("entropy_model_map.") for k in two - base) and two - base
    assert {k.split(".")[1] for k in two - base} == {"h_a", "h_s", "entropy_bottleneck"}
    bad = copy.deepcopy(syn.ABLATION_NOCONDITION_CONFIG)
    bad["g_a"]["condition_ablation"] = "something_else"
    m = pcc.ColorModel(bad)
    with pytest.raises(ValueError):
        m.g_a.condition_encoder(None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["no_condition", "not_dense", "no_condition_not_dense"])
def test_variant_codec_vs_oracle(pcc, name):
    from oracle.codec import Codec
    from _parity import compare_codec
    cfg = _configs(pcc)[name]
    syn = pcc.synthetic
    model = syn.make_model(0, DEV, config=cfg)
    model.update()
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    codec = Codec(sd, cfg)
    codec.update()
    pts = syn.sphere_shell(grid=64, radius=27.0, half_width=0.6)
    for q_g, q_a in ((0.5, 0.5), (0.1, 0.9)):
        qc, qf = syn.uniform_qmap(pts[:, :3], q_g, q_a)
        r = compare_codec(pcc, model, codec, pts, qc, qf, (name, q_g, q_a), DEV)
        assert r["bpp"] > 0
    # the q-map must matter in the ablation too (it IS the FiLM there)
    if name == "no_condition":
        x = torch.from_numpy(pts).to(DEV)
        outs = []
        for q in (0.2, 0.8):
            qc, qf = syn.uniform_qmap(pts[:, :3], q, q)
            Q = pcc.SparseTensor(coordinates=torch.from_numpy(qc).to(DEV), features=torch.from_numpy(qf).to(DEV), device=DEV)
            outs.append(model.compress(x, Q)[0])
        assert outs[0] != outs[1]
