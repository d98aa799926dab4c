"""Test helper: the package's minimal HDF5 writer (strique_amd/h5write.py)."""
from strique_amd.h5write import *      # noqa: F401,F403
from strique_amd.h5write import _File, _dataset, _group, _finish, _string_attr, _vlen_string_attr, _group_with_messages      # noqa: F401
