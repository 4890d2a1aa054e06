"""motif_amd -- MI355X-native MoTIF C-STVSR inference hot path.

HIP kernels behind a C ABI (`include/motif_hip.h`, `motif_amd/csrc`), and the host-side mirror of the
reference's Python operator surface (`motif_amd.models`, `motif_amd.OpticalFlow`).
`install_reference_namespace()` aliases the mirror under the reference's own module names so that a
reference-style driver (`from models import create_model`, `import option`) runs unchanged.
"""
import sys

__version__ = "0.1.0"


def install_reference_namespace():
    import importlib
    for ref_name, mine in (("models", "motif_amd.models"), ("option", "motif_amd.option"),
                           ("utils", "motif_amd.utils"), ("utils.util", "motif_amd.utils.util"),
                           ("OpticalFlow", "motif_amd.OpticalFlow")):
        sys.modules.setdefault(ref_name, importlib.import_module(mine))
