"""Host-side mirror of the reference's ``design_utils`` package for the hot path:
same function names, arguments and output formats; the numeric kernels run in libtimedhip.so."""
