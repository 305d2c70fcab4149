"""Import the reference's own layer classes, unmodified, behind the fake
`tensorflow` module (see fake_tensorflow.py).  BUILD-CONTAINER ONLY: the
reference tree (`/root/reference`) does not exist on the GPU box and nothing
outside `oracle/refshim/` imports this file.
"""
import importlib
import importlib.util
import os
import sys
import types

REF_ROOT = os.environ.get("AGCN_REFERENCE_ROOT", "/root/reference")

sys.dont_write_bytecode = True  # never drop __pycache__ into the read-only tree


def available():
    return os.path.isdir(os.path.join(REF_ROOT, "models", "layers"))


def load():
    """Returns (tf_shim, layers_module, graph_structure_module)."""
    here = os.path.dirname(os.path.abspath(__file__))
    if here not in sys.path:
        sys.path.insert(0, here)
    import fake_tensorflow as ftf
    ftf.install()

    # Alias packages so that `AGCN.models.layers` resolves without running
    # /root/reference/__init__.py or models/__init__.py (they import rdkit/TF trainers).
    if "AGCN" not in sys.modules:
        pkg = types.ModuleType("AGCN")
        pkg.__path__ = [REF_ROOT]
        sys.modules["AGCN"] = pkg
        mpkg = types.ModuleType("AGCN.models")
        mpkg.__path__ = [os.path.join(REF_ROOT, "models")]
        sys.modules["AGCN.models"] = mpkg
        pkg.models = mpkg
    importlib.import_module("AGCN.models.operators")
    layers = importlib.import_module("AGCN.models.layers")

    spec = importlib.util.spec_from_file_location(
        "agcn_ref_graph_structure", os.path.join(REF_ROOT, "models", "graph_structure.py"))
    gs = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gs)
    return ftf, layers, gs
