"""generalsreinforcementlearning_amd — MI355X-native batched Generals.io turn engine.

Product code: hand-written HIP kernels for gfx950 (csrc/) behind the C ABI of
include/generals_vec.h, plus this thin host mirror of the reference's
``game.Engine`` method set.  There is no CPU fallback: importing works without a
GPU (the build check), but every compute call needs the HIP library and a device.
"""
from ._lib import GvecError, lib, lib_path, load  # noqa: F401
from .vec_engine import (ACTION_DTYPE, ACT_HALF, ACT_VALID, ERR_NAMES, TILE_CITY, TILE_GENERAL, TILE_MOUNTAIN,  # noqa: F401
                         TILE_NORMAL, VecEngine, make_actions, unpack_legal_bits)

__all__ = ["VecEngine", "GvecError", "lib", "load", "lib_path", "ACTION_DTYPE", "make_actions", "unpack_legal_bits"]
