"""Import path of the reference (`from wgpu_montecarlo.transpiler import PythonToWGSL, transpile_function,
TranspilerError`, its tests/test_transpiler.py:7). The implementation lives in transpile.py / frontend.py."""
from .frontend import TranspilerError
from .transpile import PythonToHIP, PythonToWGSL, transpile_function, transpile_function_hip

__all__ = ["PythonToWGSL", "PythonToHIP", "TranspilerError", "transpile_function", "transpile_function_hip"]
