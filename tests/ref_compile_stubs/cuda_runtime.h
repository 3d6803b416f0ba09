// stub for the syntax-only compile of the reference's host code (tests/test_reference_compiles_against_boundary.py)
#pragma once
