"""The reference's gtest known-answers run against the HIP engine through the C ABI (needs a GPU)."""
import pytest

import kat_cases

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("case", kat_cases.ALL_ENGINE_CASES, ids=lambda c: c.__name__)
def test_engine_matches_reference_gtests(case, make_engine):
    case(make_engine)
