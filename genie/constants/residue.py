"""genie/constants/residue.py: the 20 residue types in the reference's order."""
from collections import OrderedDict

from genie2_amd.features import RESTYPES as _ORDER, RESTYPE_3 as _THREE

RESTYPE_1_TO_3 = OrderedDict(zip(_ORDER, _THREE))
RESTYPE_3_TO_1 = {v: k for k, v in RESTYPE_1_TO_3.items()}
RESTYPES = list(RESTYPE_1_TO_3.keys())
RESTYPE_ORDER = {restype: i for i, restype in enumerate(RESTYPES)}
