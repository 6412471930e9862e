"""Pins the CPU oracle against the reference's own known answers (no GPU)."""
import pytest

import kat_cases


def test_hash_known_answers():
    kat_cases.case_hash_known_answers()


@pytest.mark.parametrize("case", kat_cases.ALL_ENGINE_CASES, ids=lambda c: c.__name__)
def test_oracle_matches_reference_gtests(case, make_oracle):
    case(make_oracle)
