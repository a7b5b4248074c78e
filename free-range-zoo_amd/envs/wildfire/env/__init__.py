from .structures.configuration import (WildfireConfiguration, FireConfiguration, AgentConfiguration, StochasticConfiguration,
                                       RewardConfiguration)
