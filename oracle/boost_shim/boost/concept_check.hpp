#pragma once
#include <limits>
#include "boost/utility.hpp"
