"""montecarlocuda_amd -- MI355X-native Monte Carlo option pricing behind the MonteCarloCUDA host API.

The product is the C ABI in ``csrc/`` (``libmc_mi355x.so`` + the legacy-symbol libraries);
this package is the thin host-side mirror used by the tests, ``bench.py`` and Python callers.
Importing it requires the built shared library; running anything requires an MI355X.
"""
from . import _lib
from ._lib import MC_DEFAULT_SEED, McError, build
from .engine import (CVA, Engine, Estimate, MultiOptionData, OptionData, OptionValue, basket_control_mean, chol, closing,
                     dev_basketOpt, dev_cvaEquityOption, dev_vanillaOpt, factor_from_cov, pci_bus_id, shard_range)

__all__ = ["Engine", "OptionData", "MultiOptionData", "CVA", "OptionValue", "Estimate", "dev_vanillaOpt",
           "dev_basketOpt", "dev_cvaEquityOption", "basket_control_mean", "chol", "factor_from_cov", "closing", "shard_range", "pci_bus_id", "build", "McError",
           "MC_DEFAULT_SEED"]
