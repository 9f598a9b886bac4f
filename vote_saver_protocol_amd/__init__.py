"""vote_saver_protocol_amd -- MI355X-native Groth16 prover hot path (BLS12-381 MSM + Fr NTT) behind the
reference's multiexp / evaluation_domain / prover call shapes.  See DESIGN.md and include/vsp.h."""
from ._lib import SO_PATH, VspLibraryMissing, load  # noqa: F401
from . import api  # noqa: F401
from . import sharded  # noqa: F401
from .sharded import LocalExchange, ShardedMsm, TorchExchange, shard_bounds  # noqa: F401
from .api import (Bases, Context, EvaluationDomain, Keypair, ProvingKey, R1CS, VspError, fixed_base_mul, fold_jacobian, fold_jacobian_device,  # noqa: F401
                  g1_compress, g1_decompress, g2_compress, g2_decompress, groth16_prove, groth16_prove_batch, groth16_prove_batch_launch, groth16_prove_batch_finish, groth16_prove_launch, groth16_prove_finish, PackedWitness, make_evaluation_domain, multiexp,
                  multiexp_with_mixed_addition, witness_map_h, SaverPublicKey, saver_encrypt, saver_generate_keypair, saver_rerandomize,
                  fr_vector_from_blob, fr_vector_to_blob, g1_vector_from_blob, g1_vector_to_blob, proof_from_blob, vk_from_blob, vk_to_blob)
