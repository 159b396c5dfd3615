"""Drop-in mirror of the reference's `lib` package (lib/ImageOperations.py, lib/CudaOperations.py, lib/Helpers.py):
same module names, function names, arguments, return shapes and sentinels; the work runs in libmocap_hip.so."""
