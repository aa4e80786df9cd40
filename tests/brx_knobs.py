"""Test / tooling helper: a brx.Context whose A/B knobs come from the environment of the TEST process.  The library and
brotli-rs_amd/brx.py read no environment (the C ABI takes explicit arguments: brx_ctx_set_option); the suite reaches the C++-only
command loops, both builds of the assembly loop etc. by running its checks in fresh processes with these variables set."""
import os

from brotli_rs_amd import brx

# environment name -> (option, value or None = the variable's integer value)
ENV = {"BRX_DEBUG_STOP": ("command_loop", None), "BRX_LOOP_BUILD": ("loop_build", None), "BRX_NO_ORDER": ("queue_order", 0),
       "BRX_NO_DEFER": ("hand_up", 0), "BRX_PLAN_A": ("levels", 0), "BRX_PLAN_B": ("levels", 2),
       "BRX_TINY_BYTES": ("tiny_bytes", None), "BRX_NO_MIRROR": ("host_in_place", 0), "BRX_GRID_CAP": ("grid_cap", None),
       "BRX_SMALL_BYTES": ("small_bytes", None), "BRX_SMALL_WAVES": ("small_waves", None), "BRX_NO_LEVEL4": ("level4", 0)}


def options_from_env():
    opts = {}
    for env, (name, value) in ENV.items():
        if env in os.environ:
            opts[name] = int(os.environ[env]) if value is None else value
    return opts


def context(device=0, **more):
    return brx.Context(device, options=dict(options_from_env(), **more))
