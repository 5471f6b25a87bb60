"""tabmat_amd: MI355X-native sandwich / matvec / transpose_matvec hot path behind tabmat's
MatrixBase / SplitMatrix API.  Block storage lives in HBM; every product is a hand-written
HIP kernel in libtabmat_hip.so reached through the C ABI of include/tabmat_hip.h.  There is
no CPU fallback."""
from .categorical_matrix import CategoricalMatrix
from .constructor import from_csc, from_df, from_pandas
from .dense_matrix import DenseMatrix, set_strict_f32, set_strict_f64, strict_f32, strict_f64
from .matrix_base import MatrixBase
from .sparse_matrix import SparseMatrix
from .split_matrix import SplitMatrix, as_tabmat, hstack
from .standardized_mat import StandardizedMatrix

__all__ = ["DenseMatrix", "SparseMatrix", "CategoricalMatrix", "SplitMatrix",
           "StandardizedMatrix", "MatrixBase", "hstack", "as_tabmat", "from_csc", "from_df",
           "from_pandas", "set_strict_f64", "strict_f64", "set_strict_f32", "strict_f32"]
