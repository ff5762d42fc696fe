#include <thrust/execution_policy.h>
