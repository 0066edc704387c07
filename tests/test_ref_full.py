"""oracle/_ref pin of the WHOLE per-frame path: the sequences of tests/test_ref_track.py once more, this time against oracle/_ref/libref_full.so - the
reference's System / Tracking / Frame / Map / ORBextractor AND its real src/Optimizer.cc + src/Converter.cc + vendored g2o, every source compiled verbatim
from /root/reference (oracle/ref/Makefile) against the mini-cv shim, shim/Eigen and shim/cs.h.  Where test_ref_track.py has the oracle's optimisers behind the
reference's Track() (the same code on both sides of the comparison), here Optimizer::PoseOptimizationFlow2Cam / PoseOptimizationFlow2 ARE the reference's:
the BlockSolver_6_3 / 2-DoF aliasing runs through g2o's real memory layout, the Levenberg loop, the outlier rounds, the chi2 gate, the
write-backs of the refined key points and Converter's float <-> double marshalling are the reference's own statements.

What the comparison asserts is unchanged: `np.array_equal` on the pose System::TrackRGBD returns, on every renewed static / object key, correspondence, flow,
depth and 3-D point, on the per-object vectors incl. every object motion vObjMod, on the mask UpdateMask leaves behind and on every tracklet - frame by frame
over an exact sequence, a noisy one with a dropped mask, the five-box sequence (objects turning, leaving, entering; weakly constrained object problems whose
Levenberg runs 100+ iterations) and an OMD-settings sequence with sampled features.  So for rows a15 / a16 / a19 / a20 / a22 / a27-a30 (per-frame) "oracle ==
the reference's own optimiser" is shown to the last bit of what the reference hands back (CV_32F), not argued."""
import pytest

from tests import oracle_lib
from tests import test_ref_track as T


@pytest.fixture(scope="module")
def reffull():
    if oracle_lib.load_ref_full() is None:
        pytest.skip("parity unpinned: oracle/_ref/libref_full.so absent and no reference checkout to build it from")
    return True


@pytest.mark.parametrize("name", sorted(T.SEQUENCES))
def test_oracle_track_equals_the_whole_reference(oracle, reffull, name, tmp_path):
    T.run_sequence_against_the_reference(oracle, name, tmp_path, full=True)


def test_sampled_features_omd_settings_whole_reference(oracle, reffull, tmp_path):
    T.run_sampled_omd(oracle, tmp_path, full=True)
