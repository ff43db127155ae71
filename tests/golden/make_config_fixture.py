#!/usr/bin/env python
"""Dumps the flag names and defaults of the reference's ``config.get_config()`` (config.py:17-114) to
``config_defaults.json``: the drop-in surface the product's config module must reproduce.  The reference's
config.py imports only ``util.str2bool``; that one helper is supplied here (same definition as util.py:401-402) so
that the TensorFlow imports of the reference's util.py are not needed.  Run in the build container:
    python tests/golden/make_config_fixture.py
"""
import importlib.util
import json
import os
import sys
import types

u = types.ModuleType("util")
u.str2bool = lambda v: v.lower() in ("true", "1")
sys.modules["util"] = u
spec = importlib.util.spec_from_file_location("ref_config", "/root/reference/config.py")
m = importlib.util.module_from_spec(spec)
sys.argv = ["x"]
spec.loader.exec_module(m)
cfg, _ = m.get_config()
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "config_defaults.json")
json.dump({k: v for k, v in sorted(vars(cfg).items())}, open(out, "w"), indent=1)
print(out, len(vars(cfg)), "flags")
