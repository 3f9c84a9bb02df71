"""cmx — MI355X-native array evaluation of CloudMicrophysics.jl rate functions.

Host-side mirror of the reference's module layout for the hot path:
  cmx.parameters        ↔ CloudMicrophysics.Parameters (CMP)
  cmx.microphysics0m    ↔ CloudMicrophysics.Microphysics0M (CM0) + the 0M methods of BMT
  cmx.bulk_tendencies   ↔ CloudMicrophysics.BulkMicrophysicsTendencies (BMT) + per-process CM2 rates
  cmx.utilities         ↔ CloudMicrophysics.Utilities (UT): gamma_inc / gamma_inc_inv over columns
  cmx.distribution_tools ↔ CloudMicrophysics.DistributionTools (DT) + the SB2006 PSD accessors of CM2
  cmx.synthetic         ↔ the state generator of test/gpu_performance.jl:80-136
  cmx.sharding          ↔ (no reference equivalent) one-process-per-GPU sharding + RCCL diagnostic sums
All compute goes through libcmx.so (include/cmx.h); there is no CPU fallback.
"""
from . import _abi, parameters, sharding, synthetic  # noqa: F401
from ._lib import CmxLibraryError, CmxStatusError  # noqa: F401
from .bulk_tendencies import (Chen2022VelTypeRain, Microphysics2Moment, SB2006ProcessRates,  # noqa: F401
                              SB2006VelType, Tendencies2MP3, WarmRainTendencies2M, bulk_microphysics_tendencies,
                              bulk_microphysics_tendencies_fields, bulk_microphysics_tendencies_2m_p3_fields,
                              bulk_2m_cloud_to_rain, cloud_terminal_velocity, column_sums, sb2006_process_rates,
                              ColumnTendencies2M, column_tendencies_sedimentation)

from .ice_nucleation import (IceNucleationRates, a_w_eT, a_w_ice, domain_error_count,  # noqa: F401
                             ice_nucleation_rates, liquid_freezing_rate, h2so4_solution, mohler2006_deposition, MohlerDeposition, deposition_J,
                             INP_concentration_frequency)

from .microphysics1m import (Instantaneous, LinearizedAverage, Microphysics1Moment, SedimentationVelocities, SourceTerms1M, Tendencies1M,  # noqa: F401
                             TerminalVelocities1M, bulk_microphysics_tendencies_1m, bulk_microphysics_tendencies_1m_fields,
                             microphysics_source_terms_1m, sedimentation_velocities, terminal_velocity_1m, ColumnStep1M,
                             column_tendencies_sedimentation_1m)

from .microphysics0m import (Microphysics0Moment, bulk_microphysics_tendencies_0m, d_remove_precipitation_d_q_tot,  # noqa: F401
                             remove_precipitation)

from .aerosol import (ActivationResult, AerosolDistribution, ModeColumns, Mode_B, Mode_kappa, aerosol_activation,  # noqa: F401
                      aerosol_activation_columns, total_activated)

from .p3 import (P3Melt, P3Shape, P3ShapeVelocities, P3Velocities, p3_shape_and_terminal_velocities, p3_het_ice_nucleation, p3_ice_melt, p3_ice_self_collection, p3_liquid_ice_collisions,  # noqa: F401
                 p3_shape, p3_terminal_velocities)

from .utilities import GammaInc, gamma_inc, gamma_inc_inv  # noqa: F401
from . import cloud_diagnostics  # noqa: F401
from .distribution_tools import Distribution, SizeDistribution, exponential_distribution, generalized_gamma, size_distribution  # noqa: F401

__version__ = "0.1.0"
