"""Import-name alias so ``from sageattention import sageattn`` (every example of the reference,
e.g. example/cogvideox_infer.py:34-35) resolves to the gfx950 implementation unchanged."""
from sageattention_amd import *          # noqa: F401,F403
from sageattention_amd import __all__, __version__   # noqa: F401
