"""`python genie/train.py -c <config>`: the reference's entry point (genie/train.py:70-81)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from genie2_amd.train import cli, main  # noqa: E402,F401

if __name__ == '__main__':
    cli()
