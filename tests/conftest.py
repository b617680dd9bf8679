import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    # The libraries are built in-tree (and git-ignored): build them when a fresh checkout has none.
    libs = [os.path.join(ROOT, "functracer_amd", "lib", "libfunctracer_hip.so"), os.path.join(ROOT, "functracer_amd", "lib", "libfunctracer_host.so"),
            os.path.join(ROOT, "oracle", "libft_oracle.so")]
    if not all(os.path.exists(p) for p in libs):
        import __graft_entry__
        __graft_entry__.build()


@pytest.fixture(scope="session")
def golden():
    import json
    with open(os.path.join(ROOT, "tests", "golden", "known_answers.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def hip():
    """One device context for the whole session.  No skip and no fallback: on a GPU box a missing
    library or device is a failure."""
    import functracer_amd as ft
    ctx = ft.Context(device=0)
    yield ctx
    ctx.close()


def scene_path(name):
    return os.path.join(ROOT, "scenes", name + ".scene")
