"""ssvio_amd -- MI355X (gfx950) ORB extraction + brute-force Hamming matching behind
SSVIO's ORBextractor / FeatureMatching interfaces.  See DESIGN.md."""
from ._lib import KP_DTYPE, DM_DTYPE, SsvioError, device_count, MODE_BRUTEFORCE, MODE_KNN2_RATIO, MODE_CROSSCHECK
from .extractor import ORBextractor, distribute_octtree
from .matching import FeatureMatching
from .vocabulary import Vocabulary

__all__ = ["ORBextractor", "FeatureMatching", "Vocabulary", "distribute_octtree", "KP_DTYPE", "DM_DTYPE", "SsvioError",
           "device_count", "MODE_BRUTEFORCE", "MODE_KNN2_RATIO", "MODE_CROSSCHECK"]
