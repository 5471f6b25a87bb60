"""Host-side mirror of tabmat's Cython extension modules (src/tabmat/ext/*.pyx in the
reference): same function names and argument meaning, but every array is a device
buffer and the work is done by libtabmat_hip.so through the C ABI."""
