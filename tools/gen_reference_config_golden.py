"""Write tests/golden/reference_base_config.json: the reference's shipped defaults (config/base.yaml, read as DATA where it lies) with the
values params.cpp derives from them (read_base_params, params.cpp:330-443): reset_every = reset_alpha_every * refine_every,
vis_batch_pt_num = 50 * batch_pt_num, center_reg = 0 (the key is absent: cv::FileNode >> bool leaves the default), numerical_grad forced on
for the tcnn decoder.  tests/test_reference_config_defaults.py holds this repository's configuration defaults to it."""
import json
import os

import yaml

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = "/root/reference/config/base.yaml"
text = open(src).read().replace("%YAML:1.0", "", 1)          # OpenCV's directive is not YAML 1.1 syntax
cfg = yaml.safe_load(text)
cfg = {k: (float(v) if isinstance(v, str) else v) for k, v in cfg.items()}    # "1e-1" style scalars load as strings in YAML 1.1
derived = dict(reset_every=int(cfg["reset_alpha_every"]) * int(cfg["refine_every"]), vis_batch_pt_num=50 * int(cfg["batch_pt_num"]),
               center_reg=int(cfg.get("center_reg", 0)), numerical_grad_effective={"decoder_implementation_0": int(cfg["numerical_grad"]), "decoder_implementation_1": 1})
out = {"source": "config/base.yaml of the reference (+ params.cpp:330-443 for the derived values)", "base": cfg, "derived": derived}
path = os.path.join(ROOT, "tests", "golden", "reference_base_config.json")
json.dump(out, open(path, "w"), indent=1, sort_keys=True)
print(path, len(cfg), "keys")
