import torch

from distributed_machine_learning_amd.models import (VGG11, count_parameters, expected_keys,
                                                     layer_plan, summary_string)
from distributed_machine_learning_amd.models.checkpoint import (expected_shapes, load_checkpoint,
                                                                save_checkpoint,
                                                                validate_state_dict)


def test_parameter_counts():
    assert count_parameters("VGG11", False) == 9_225_610
    assert count_parameters("VGG11", True) == 9_231_114
    m = VGG11()
    assert sum(p.numel() for p in m.parameters()) == 9_225_610
    assert len(list(m.parameters())) == 18
    assert len(list(VGG11(True).parameters())) == 34


def test_state_dict_keys_match_reference_layout():
    nobn = list(VGG11().state_dict().keys())
    assert nobn == expected_keys("VGG11", False)
    assert nobn[:2] == ["layers.0.weight", "layers.0.bias"]
    assert [k for k in nobn if k.endswith("weight")] == \
        [f"layers.{i}.weight" for i in (0, 3, 6, 8, 11, 13, 16, 18)] + ["fc1.weight"]
    bn = list(VGG11(True).state_dict().keys())
    assert bn == expected_keys("VGG11", True)
    assert "layers.1.running_mean" in bn and "layers.26.num_batches_tracked" in bn
    assert tuple(VGG11().state_dict()["fc1.weight"].shape) == (10, 512)


def test_forward_shape_all_archs():
    from distributed_machine_learning_amd.models import VGG
    x = torch.randn(2, 3, 32, 32)
    for arch in ("VGG11", "VGG13", "VGG16", "VGG19"):
        assert VGG(arch)(x).shape == (2, 10)


def test_summary_matches_published_table():
    s = summary_string("VGG11", False)
    assert "Total params: 9,225,610" in s
    assert "Params size (MB): 35.19" in s
    assert "Estimated Total Size (MB): 37.76" in s
    assert "Conv2d-1" in s and "Linear-22" in s
    assert "[-1, 64, 32, 32]" in s
    assert "Linear-30" in summary_string("VGG11", True)


def test_checkpoint_roundtrip_and_ddp_prefix(tmp_path):
    m = VGG11(True)
    sd = m.state_dict()
    validate_state_dict(sd, "VGG11", True)
    p = str(tmp_path / "ck.pt")
    save_checkpoint(p, sd, ddp_prefix=True)
    raw = torch.load(p, weights_only=True)
    assert all(k.startswith("module.") for k in raw)
    back = load_checkpoint(p)
    assert list(back.keys()) == list(sd.keys())
    m2 = VGG11(True)
    m2.load_state_dict(back)
    for k in sd:
        assert torch.equal(m2.state_dict()[k], sd[k])
    assert expected_shapes("VGG11", False)["layers.18.weight"] == (512, 512, 3, 3)


def test_native_module_has_the_reference_state_dict():
    from distributed_machine_learning_amd.models import VGG
    from distributed_machine_learning_amd.models import NativeVGG
    for bn in (False, True):
        torch.manual_seed(5)
        a = VGG("VGG11", bn)
        torch.manual_seed(5)
        b = NativeVGG("VGG11", bn)
        sa, sb = a.state_dict(), b.state_dict()
        assert list(sa.keys()) == list(sb.keys())
        for k in sa:
            assert sa[k].shape == sb[k].shape and torch.equal(sa[k], sb[k]), k
