// placeholder: replaced by the MuJoCo gym kernels
#include "engine.h"
namespace epa {
bool DescribeMujoco(const std::string&, const Config&, std::vector<KeySpec>*, KeySpec*) { return false; }
Pool* MakeMujoco(const std::string&, const Config&) { return nullptr; }
}
