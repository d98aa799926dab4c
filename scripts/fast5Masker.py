#!/usr/bin/env python3
"""Drop-in for the reference's scripts/fast5Masker.py (host code; see strique_amd/masker.py)."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from strique_amd.masker import main  # noqa: E402

if __name__ == "__main__":
    main()
