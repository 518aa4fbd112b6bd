import importlib.util
import os

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
_spec = importlib.util.spec_from_file_location("suite_conftest", os.path.join(_root, "tests", "conftest.py"))
_suite = importlib.util.module_from_spec(_spec)
_spec.loader.exec_module(_suite)
oracle, golden = _suite.oracle, _suite.golden  # the suite's own fixtures
