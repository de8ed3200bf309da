import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
for p in (ROOT, GOLDEN):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_sessionstart(session):
    """A fresh checkout has no built artefacts (they are git-ignored): build them once, so that the suite does
    not depend on somebody having called __graft_entry__.build() first.  hipcc cross-compiles without a GPU."""
    need = [os.path.join(ROOT, "eval_driving_safety_amd", "libadvengine.so"), os.path.join(ROOT, "eval_driving_safety_amd", "libadvengine_hooks.so"),
            os.path.join(ROOT, "oracle", "_build", "liboracle.so")]
    if all(os.path.exists(p) for p in need):
        return
    try:
        import __graft_entry__
        __graft_entry__.build()
    except Exception as e:          # leave it to the tests to report what is missing
        print("conftest: build() failed: %r" % (e,), file=sys.stderr)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: long repeats of coverage the default suite already has; run with --full or ADV_FULL_SUITE=1")


def pytest_addoption(parser):
    parser.addoption("--full", action="store_true", default=False, help="also run the tests marked slow (fuzz repeats, second-opinion comparisons)")


def pytest_collection_modifyitems(config, items):
    if config.getoption("--full") or os.environ.get("ADV_FULL_SUITE") == "1":
        return
    skip = pytest.mark.skip(reason="slow: run with --full or ADV_FULL_SUITE=1 (the default -m gpu suite must stay well inside the driver's step limit)")
    for item in items:
        if "slow" in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def route(monkeypatch):
    """``with route(ADV_CONV_NO_DMA="1"): ...`` - inside the block the calls go to libadvengine_hooks.so (the -DADV_TEST_HOOKS build of
    the same sources) with the given route switches in the environment; outside it to the shipped library, which reads none."""
    import contextlib

    @contextlib.contextmanager
    def cm(**env):
        from eval_driving_safety_amd import _lib
        with monkeypatch.context() as m, _lib.using(_lib.HOOKS_LIB_PATH) as lib:
            assert lib.adv_build_has_test_hooks() == 1
            for k in [k for k in os.environ if k.startswith("ADV_CONV_") or k.startswith("ADV_ROI_")]:
                m.delenv(k)
            for k, v in env.items():
                m.setenv(k, v)
            yield

    return cm


@pytest.fixture
def checkout(request):
    """``checkout("dsgn_checkout", ext="reference")``: put one stand-in checkout on sys.path for the test (DSGN: with the given module at
    ``dsgn._C``, see tests/_upstream.py) and forget its modules afterwards"""
    import _upstream

    def use(name, ext="reference"):
        path = os.path.join(_upstream.FAKE, name)
        sys.path.insert(0, path)
        request.addfinalizer(lambda: sys.path.remove(path))
        if name == "dsgn_checkout":
            _upstream.bind_dsgn_extension(ext)
        return path
    yield use
    from eval_driving_safety_amd import adopt_functional
    adopt_functional.unbind()
    _upstream.forget_upstream()


@pytest.fixture(scope="session")
def golden_index():
    with open(os.path.join(GOLDEN, "index.json")) as f:
        return json.load(f)


@pytest.fixture(scope="session")
def golden():
    cache = {}

    def load(name):
        if name not in cache:
            with np.load(os.path.join(GOLDEN, name + ".npz")) as z:
                cache[name] = {k: z[k] for k in z.files}
        return cache[name]

    return load
