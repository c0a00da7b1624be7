"""kmdiff_amd -- MI355X-native hot path of `kmdiff diff` (tlemane/kmdiff).

Host-side mirror of the reference's operator interface for the path (names follow the
reference: PoissonLikelihood, diff_observer, make_corrector / aggregator) over the C-ABI of
libkmdiff_hip.so.  All arithmetic runs in the HIP library; nothing here computes results on
the CPU.
"""
from ._native import (KmdError, LIB_PATH, SIGN_CONTROL, SIGN_CASE, SIGN_NO, CORR_NOTHING,
                      CORR_BONFERRONI, CORR_BENJAMINI, CORR_SIDAK, CORR_HOLM, LAYOUT_ROWS,
                      LAYOUT_SOA, LAYOUT_TILED)
from .hip import (DeviceBuffer, PoissonLikelihood, CountMatrix, SurvivorAccumulator,
                  diff_observer, aggregate, synth_matrix, column_sums, device_count,
                  pop_strat_corrector, gather_counts, merge_partition, merge_sums, merge_filter, merge_filter_batch, StreamSet, RowSums, gather_counts_streams, PopulationPCA, pca_eigen, synth_streams, pack_streams, unpack_streams,
                  device_name, Event, CORRECTION_BY_NAME, SYNTH_MIXED)

__all__ = ["KmdError", "LIB_PATH", "DeviceBuffer", "PoissonLikelihood", "CountMatrix",
           "SurvivorAccumulator", "diff_observer", "aggregate", "pop_strat_corrector", "gather_counts", "merge_partition", "merge_sums", "merge_filter", "merge_filter_batch", "StreamSet", "RowSums", "gather_counts_streams", "PopulationPCA", "pca_eigen", "synth_matrix", "synth_streams", "pack_streams", "unpack_streams", "column_sums",
           "device_count", "device_name", "Event", "CORRECTION_BY_NAME", "SYNTH_MIXED",
           "SIGN_CONTROL", "SIGN_CASE", "SIGN_NO", "CORR_NOTHING", "CORR_BONFERRONI",
           "CORR_BENJAMINI", "CORR_SIDAK", "CORR_HOLM", "LAYOUT_ROWS", "LAYOUT_SOA", "LAYOUT_TILED"]
