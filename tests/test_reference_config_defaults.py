"""This repository's configuration defaults against the reference's shipped ones (CPU test): tests/golden/reference_base_config.json is
config/base.yaml of the reference (read as data by tools/gen_reference_config_golden.py) plus the values params.cpp:330-443 derives.
A user who switches from the reference and keeps the defaults trains with the reference's loss weights, learning rates, refinement
thresholds and grid / decoder shape: GSConfig (Python mirror and gsdf_model::), MapConfig, gsdf_extras::JointConfig (what bench.py times),
the SDF-side defaults of gs_sdf_amd.sdf.LocalMap, the splat parameter groups' learning rates (neural_gaussian.cpp:434-453)."""
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_base_config.json")))
B, D = GOLD["base"], GOLD["derived"]
GS_KEYS = ("near", "far", "use_absgrad", "prune_opa", "grow_grad2d", "grow_scale3d", "grow_scale2d", "prune_scale3d", "refine_scale2d_stop_iter",
           "refine_start_iter", "refine_every", "sh_degree_interval", "lr_end", "detach_sdf_grad")


def same(a, b):
    return bool(a) == bool(b) if isinstance(a, bool) or isinstance(b, bool) else a == pytest.approx(b, rel=1e-6)


def test_fixture_is_the_references_file():
    src = "/root/reference/config/base.yaml"
    if not os.path.exists(src):
        pytest.skip("the reference tree is only present in the build container")
    import yaml
    cfg = yaml.safe_load(open(src).read().replace("%YAML:1.0", "", 1))
    assert set(cfg) == set(B)
    for k, v in cfg.items():
        assert float(v) == pytest.approx(float(B[k])), k
    assert "center_reg" not in cfg                      # absent -> k_center_reg = 0: stochastic samples are the reference's default


def test_python_mirror_defaults():
    from gs_sdf_amd.neural_gs import GSConfig
    c = GSConfig()
    for k in GS_KEYS:
        assert same(getattr(c, k), B[k]), k
    assert c.reset_every == D["reset_every"] and bool(c.center_reg) == bool(D["center_reg"])
    assert c.pause_refine_after_reset == 0 and int(B["pause_refine"]) == 0      # neural_gaussian.cpp:288-292


def test_cpp_model_and_joint_step_defaults():
    import gs_sdf_amd.hostlib as hl
    host = hl.load()
    g = host.GSConfig()
    for k in GS_KEYS:
        assert same(getattr(g, k), B[k]), k
    assert g.reset_every == D["reset_every"] and g.vis_batch_pt_num == D["vis_batch_pt_num"]
    assert bool(g.center_reg) == bool(D["center_reg"]) and bool(g.geo_init) == bool(B["geo_init"])
    m = host.MapConfig()
    for k in ("n_levels", "n_features_per_level", "log2_hashmap_size", "hidden_dim", "geo_num_layer", "free_sample_num", "decoder_implementation"):
        assert same(getattr(m, k), B[k]), k
    assert m.base_resolution == 32 and m.per_level_scale == 2.0            # fixed in encoding_map.cpp:15-23, not in the yaml
    j = host.joint_config_defaults()
    for k in ("near", "far", "rgb_weight", "dssim_weight", "eikonal_weight", "gs_sdf_weight", "visible_thr", "sdf_weight", "align_weight",
              "render_normal_weight", "isotropic_weight", "lr_end"):
        assert same(j[k], B[k]), k
    assert j["analytic"] == (D["numerical_grad_effective"]["decoder_implementation_0"] == 0) and j["reference_terms"]
    assert j["lrs"] == pytest.approx([1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 2.5e-3 / 20])     # neural_gaussian.cpp:434-453 (x spatial_scale for offsets)


def test_python_splat_groups_and_sdf_defaults():
    import inspect
    import torch
    from gs_sdf_amd.neural_gs import GSConfig, NeuralGS
    import gs_sdf_amd.sdf as sdf
    n = 4
    gs = NeuralGS(torch.zeros(n, 3), torch.zeros(n, 3), torch.zeros(n, 4), torch.zeros(n), torch.zeros(n, 1, 3), torch.zeros(n, 0, 3), GSConfig())
    assert [g["lr"] for g in gs.param_groups()] == pytest.approx([1.6e-4, 5e-3, 1e-3, 5e-2, 2.5e-3, 2.5e-3 / 20])
    assert all(g["eps"] == 1e-15 for g in gs.param_groups())
    sig = inspect.signature(sdf.LocalMap.__init__).parameters
    assert sig["decoder_implementation"].default == B["decoder_implementation"]
    for name, key in (("hidden_dim", "hidden_dim"), ("geo_num_layer", "geo_num_layer")):
        if name in sig:
            assert sig[name].default == B[key], name
