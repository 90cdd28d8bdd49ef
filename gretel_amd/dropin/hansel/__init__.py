"""Importable `hansel` module for an unmodified Gretel: put `gretel_amd/dropin` on PYTHONPATH and
the reference's `from hansel import Hansel` (gretel/gretel.py:7, gretel/util.py:4) resolves to the
device-backed class.  See INTEGRATION.md."""
from gretel_amd.hansel import Hansel, HanselSymbol  # noqa: F401

__version__ = "0.0.92+mi355x"
