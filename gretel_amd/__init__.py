"""gretel_amd -- Gretel's hot path (Hansel tensor, BAM->Hansel fill, L'th-order Markov path
extension, reweighting) as hand-written HIP kernels for MI355X behind the reference's Python API.
See DESIGN.md / INTEGRATION.md."""
__version__ = "0.1.0"
