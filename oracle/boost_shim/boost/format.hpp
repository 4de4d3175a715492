// Stand-in for boost/format.hpp: included by the reference's option parser but not used there.
#pragma once
