"""Public names of the package (re-exported by the ``stark_symphony_amd`` import shim)."""
from .formats import (MalformedProof, Stark101Proof, StwoConfig, StwoProof,  # noqa: F401
                      PRODUCTION_CONFIG, TESTING_CONFIG, stark101_from_json, stark101_from_simf,
                      stark101_from_transcript, stark101_from_wit, stark101_to_json, stark101_to_simf, stark101_to_wit,
                      stwo_from_json, stwo_from_simf, stwo_from_wit, stwo_to_json, stwo_to_simf,
                      stwo_to_wit)

__all__ = ["MalformedProof", "Stark101Proof", "StwoConfig", "StwoProof", "PRODUCTION_CONFIG",
           "TESTING_CONFIG", "stark101_from_json", "stark101_from_simf", "stark101_from_transcript", "stark101_from_wit",
           "stark101_to_json", "stark101_to_simf", "stark101_to_wit", "stwo_from_json",
           "stwo_from_simf", "stwo_from_wit", "stwo_to_json", "stwo_to_simf", "stwo_to_wit"]
