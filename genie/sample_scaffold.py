"""python genie/sample_scaffold.py --name ... (same flags as the reference)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd.sample_scaffold import ScaffoldRunner, build_parser, main  # noqa: E402,F401

if __name__ == '__main__':
    main(build_parser().parse_args())
